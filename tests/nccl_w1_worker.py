"""Worker of test_hip_accelerator.test_collectives_through_rccl_in_a_group_of_one (run as a child process with a time limit, so that
an RCCL bring-up problem cannot hang the test process).  Four runs of the same three pre-training steps on cuda:0:
  A  no process group (the plain W = 1 path);
  B  backend 'nccl' (RCCL), world_size 1, FORCE_COLLECTIVES: agreement MAX all-reduce, ITC all_gather with slice backward,
     arena broadcast, ReduceOp.AVG all-reduces issued from the tower hooks / the ViT chunk hand-over on the communication
     stream, fp32 exchange;
  C  the same with the bf16 wire format (pack -> all-reduce -> unpack on the communication stream);
  D  as B with XFM_DP_NATIVE: the gradient ranges leave through the library's own communicator (xfm_dp_bucket_allreduce, on the
     communication stream itself) instead of ProcessGroupNCCL.
Mean over one rank is the identity: every range that leaves through RCCL -- from the tower hooks and the ViT chunk hand-over inside
backward, or from the sweep after it -- must come back BIT FOR BIT (fp32) or as exactly its bf16 rounding (bf16 wire format); the
worker snapshots each range on the communication stream right before its collective and compares after the step.  Run-to-run the
gradients themselves differ in the last bits (float atomics in the small-problem weight-gradient path), so A vs B is held to that
noise on the first step's gradients, not bitwise.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from golden_util import load, state_from_spec  # noqa: E402
from xfm_amd import pretrain_loop as PL  # noqa: E402
from xfm_amd import synthetic as syn  # noqa: E402


def run(meta, force, exchange="fp32", native=False):
    from xfm_amd.accelerators import RCCLDDPAccelerator
    from xfm_amd.accelerators import rccl_ddp_accelerator as A
    A._NATIVE = bool(native)   # (the module reads XFM_DP_NATIVE at import; the worker switches it per run)
    from xfm_amd.model_pretrain import XFM
    cfg = {"use_beit_v2": True, "image_res": 224, "patch_size": 16, "local_attn_depth": -1, "text_encoder": "roberta-base",
           "text_num_hidden_layers": meta["text_layers"], "text_fusion_start_at": meta["text_layers"],
           "fusion_num_hidden_layers": meta["fusion_layers"], "fusion_fusion_start_at": 0, "embed_dim": 256, "temp": 0.07,
           "learnable_temp": True, "max_temp": 0.5, "min_temp": 0.001, "vision_depth": 12}   # 12 blocks: the DEFAULT chunk of four hands over twice
    m = XFM(cfg)
    sd = syn.formula_state_dict(m.state_dict())
    m.load_state_dict(sd, strict=True)
    m.cuda()
    opt = PL.create_optimizer(PL.AttrDict(lr=1e-3, weight_decay=0.05, lr_mult=2), m)
    acc = RCCLDDPAccelerator({"RNG_SEED": 3, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1, "FORCE_COLLECTIVES": force,
                              "GRAD_EXCHANGE_DTYPE": exchange})
    wrapped, opt, _ = acc.set_up(m, opt, None, 0, 1, 0)
    m.eval()
    B = meta["B"]
    stats, grads = [], None
    snaps, identity = [], {"ranges": 0, "bad": 0}
    if force:
        # Poison the ordering the exchange must respect: the LM head's vocabulary weight gradient (50265 x 768, inside the fusion tower's
        # arena range, launched on the weight-gradient side stream and re-joined only at the end of the backward pass) is held back by a
        # ~10 ms spin on that stream.  An exchange of the fusion range that does not wait for the side stream snapshots the range
        # before the gradient lands, and the identity check below fails deterministically.
        from xfm_amd import xroberta as XR
        orig_tn = XR._WgradStream.gemm_tn

        def slow_tn(self, dy, x, dw, **kw):
            if self.on and dw.shape[0] >= 50000:
                with torch.cuda.stream(self.side):
                    torch.cuda._sleep(20_000_000)
            return orig_tn(self, dy, x, dw, **kw)

        XR._WgradStream.gemm_tn = slow_tn
        orig = acc._exchange

        def snapshotting(a, b, async_ok=True):   # runs under torch.cuda.stream(comm stream): the clone is ordered before the collective
            snaps.append((a, b, m._arena.grad[a:b].clone()))
            return orig(a, b, async_ok)

        acc._exchange = snapshotting
    first_grads = None
    for step in range(3):
        b = {k: v.cuda() for k, v in syn.pretrain_batch(B, seed=100 + step).items()}
        masks = syn.mim_block_mask(B, 14, 75, seed=100 + step)
        losses = wrapped(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                         masked_ids=b["masked_ids"], ret_mim_loss=True, data_source="image", ids_mask=masks,
                         neg_idx=([(i + 1) % B for i in range(B)], [(i + 2) % B for i in range(B)]))
        acc.backward_step(losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"], opt)
        torch.cuda.synchronize()
        for a_, b_, pre in snaps:   # what came back from RCCL vs what went in
            post = m._arena.grad[a_:b_]
            want = pre.to(torch.bfloat16).float() if exchange == "bf16" else pre
            identity["ranges"] += 1
            identity["bad"] += int(not torch.equal(post, want))
        snaps.clear()
        if step == 0:
            first_grads = m._arena.grad.clone()
        stats.append(dict(acc.stats, overlapped_ranges=len(getattr(acc, "overlapped_ranges", []))))
        if step == 2:
            grads = m._arena.grad.clone()
        acc.optimizer_step(opt, m)
    torch.cuda.synchronize()
    if force:
        XR._WgradStream.gemm_tn = orig_tn
    layout = [(m._arena.names[id(p)],) + tuple(m._arena.offsets[id(p)]) for p in m._arena.params]
    return m._arena.data.clone(), first_grads, stats, str(acc._op), sum(b_ - a for a, b_ in acc.live_ranges()), identity, layout


def diff_table(ga, gb, layout, top=25):
    """Per-parameter rel-L2 of two gradient arenas, largest first: which tensors moved says which kernel produced the odd values."""
    rows = []
    for name, o, n in layout:
        a, b = ga[o:o + n].double(), gb[o:o + n].double()
        na = float(a.norm())
        if na > 0:
            rows.append((float((a - b).norm()) / na, name, na))
    rows.sort(reverse=True)
    lines = [f"  {r[0]:.3e}  {r[1]}  |g| = {r[2]:.3e}" for r in rows[:top]]
    lines.append(f"  tensors differing by more than 1e-6: {sum(1 for r in rows if r[0] > 1e-6)} of {len(rows)}")
    return "\n".join(lines)


def main():
    torch.cuda.set_device(0)
    _, meta = load("pretrain_small")
    pa, ga, _, _, live, _, layout = run(meta, False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29653")
    dist.init_process_group("nccl", world_size=1, rank=0)
    pb, gb, sb, op, _, id_fp32, _ = run(meta, True)
    pc, gc, sc, _, _, id_bf16, _ = run(meta, True, "bf16")
    pd, gd, sd, _, _, id_native, _ = run(meta, True, "fp32", native=True)   # the gradient ranges through the C-ABI's own communicator
    dist.barrier()
    dist.destroy_process_group()
    out = {"backend": "nccl", "op": op, "live_elems": live,
           "identity_fp32": id_fp32, "identity_bf16": id_bf16, "identity_native": id_native,
           "first_step_grad_rel_l2_native": float((gd - ga).norm() / ga.norm()),
           "first_step_grad_rel_l2_vs_no_collectives": float((gb - ga).norm() / ga.norm()),
           "first_step_grad_rel_l2_bf16_wire": float((gc - ga).norm() / ga.norm()),
           "param_rel_l2_after_3_steps": float((pb - pa).norm() / pa.norm()),
           "stats_fp32": sb, "stats_bf16": sc}
    if out["first_step_grad_rel_l2_vs_no_collectives"] > 1e-6:   # a recurrence names its tensors
        print("first-step gradients, run without collectives vs run with (per parameter, rel-L2):\n" + diff_table(ga, gb, layout), flush=True)
    print("NCCL_W1 " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
