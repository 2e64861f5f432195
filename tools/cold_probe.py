"""Localise a run-to-run divergence of the FIRST step of a cold process.

  python tools/cold_probe.py --runs 24            parent: N cold child processes, one step each, per-parameter outlier table
  python tools/cold_probe.py --child [--poison]    one cold process: first step of the small pre-training model -> one JSON line

Every child builds the model of tests/nccl_w1_worker.py (no process group), runs forward + backward of step 0 and prints, per
parameter, the gradient's l2 norm and its projection on a fixed +-1 vector (two runs that differ anywhere in the tensor differ in the
projection), plus the four losses bit for bit.  The parent takes the per-parameter median over the runs as the reference and lists
every (run, parameter) whose projection is further than 1e-5 * |g| from it: the SET of tensors that moved together says which
kernel / stream hand-over produced the odd values (one dW alone: its weight-gradient launch; every tensor below some layer: an
activation gradient at that layer; the losses too: the forward).

--poison: before the step the child fills several GB of device memory with NaN bit patterns and returns it to torch's caching
allocator, so every buffer the step allocates starts as NaNs instead of the zeros of fresh pages or the previous run's (plausible)
values: a kernel that reads a buffer before its producer has written it -- a missing stream dependency -- then shows as NaN instead of
hiding behind stale data."""
import argparse
import json
import os
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(poison, steps):
    import torch
    from golden_util import load
    from xfm_amd import pretrain_loop as PL
    from xfm_amd import synthetic as syn
    from xfm_amd.accelerators import RCCLDDPAccelerator
    from xfm_amd.model_pretrain import XFM
    torch.cuda.set_device(0)
    _, meta = load("pretrain_small")
    cfg = {"use_beit_v2": True, "image_res": 224, "patch_size": 16, "local_attn_depth": -1, "text_encoder": "roberta-base",
           "text_num_hidden_layers": meta["text_layers"], "text_fusion_start_at": meta["text_layers"],
           "fusion_num_hidden_layers": meta["fusion_layers"], "fusion_fusion_start_at": 0, "embed_dim": 256, "temp": 0.07,
           "learnable_temp": True, "max_temp": 0.5, "min_temp": 0.001, "vision_depth": 6}
    m = XFM(cfg)
    m.load_state_dict(syn.formula_state_dict(m.state_dict()), strict=True)
    m.cuda()
    opt = PL.create_optimizer(PL.AttrDict(lr=1e-3, weight_decay=0.05, lr_mult=2), m)
    acc = RCCLDDPAccelerator({"RNG_SEED": 3, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1, "FORCE_COLLECTIVES": False})
    wrapped, opt, _ = acc.set_up(m, opt, None, 0, 1, 0)
    m.eval()
    B = meta["B"]
    if poison:
        torch.cuda.synchronize()
        blocks = []
        for mb in (1, 2, 3, 5, 8, 12, 20, 32, 48, 64, 96, 128, 192, 256, 384, 512) * 3:
            t = torch.empty(mb * 262144, dtype=torch.int32, device="cuda")
            t.fill_(-1)          # 0xFFFFFFFF: NaN as fp32, as two bf16, and an absurd index
            blocks.append(t)
        torch.cuda.synchronize()
        del blocks, t
    out = {"poison": bool(poison), "steps": []}
    arena = m._arena
    gen = torch.Generator(device="cpu").manual_seed(1)
    signs = {}
    for step in range(steps):
        b = {k: v.cuda() for k, v in syn.pretrain_batch(B, seed=100 + step).items()}
        masks = syn.mim_block_mask(B, 14, 75, seed=100 + step)
        losses = wrapped(b["image"], b["text_ids"], b["text_atts"], text_ids_masked=b["text_ids_masked"], masked_pos=b["masked_pos"],
                         masked_ids=b["masked_ids"], ret_mim_loss=True, data_source="image", ids_mask=masks,
                         neg_idx=([(i + 1) % B for i in range(B)], [(i + 2) % B for i in range(B)]))
        acc.backward_step(losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"], opt)
        torch.cuda.synchronize()
        rec = {"losses": {k: struct.pack(">f", float(losses[k])).hex() for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")},
               "loss_values": {k: float(losses[k]) for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")}, "params": {}}
        for p in arena.params:
            name = arena.names[id(p)]
            o, n = arena.offsets[id(p)]
            g = arena.grad[o:o + n].double()
            if name not in signs:
                signs[name] = (torch.randint(0, 2, (n,), generator=gen, dtype=torch.int8).double() * 2 - 1).cuda()
            l2 = float(g.norm())
            if l2 != l2 or l2 > 0:
                rec["params"][name] = [l2, float((g * signs[name]).sum()), int(torch.isnan(g).sum())]
        out["steps"].append(rec)
        acc.optimizer_step(opt, m)
    torch.cuda.synchronize()
    print("COLD_PROBE " + json.dumps(out), flush=True)


def parent(runs, poison, steps, tol):
    recs = []
    for r in range(runs):
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--steps", str(steps)] + (["--poison"] if poison else [])
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [l for l in p.stdout.splitlines() if l.startswith("COLD_PROBE ")]
        if p.returncode != 0 or not lines:
            print(f"run {r}: rc {p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-3000:]}", flush=True)
            continue
        recs.append(json.loads(lines[-1][len("COLD_PROBE "):]))
        print(f"run {r}: losses step0 {recs[-1]['steps'][0]['loss_values']}", flush=True)
    import statistics
    summary = {"runs": len(recs), "poison": poison, "odd": []}
    for s in range(steps):
        names = list(recs[0]["steps"][s]["params"].keys())
        ref_loss = recs[0]["steps"][s]["losses"]
        for r, rec in enumerate(recs):
            if rec["steps"][s]["losses"] != ref_loss:
                print(f"step {s} run {r}: LOSS BITS differ from run 0: {rec['steps'][s]['loss_values']} vs {recs[0]['steps'][s]['loss_values']}")
                summary["odd"].append({"step": s, "run": r, "loss": rec["steps"][s]["loss_values"]})
        for name in names:
            vals = [rec["steps"][s]["params"].get(name, [0.0, 0.0, 0]) for rec in recs]
            med_l2 = statistics.median(v[0] for v in vals)
            med_pr = statistics.median(v[1] for v in vals)
            for r, v in enumerate(vals):
                dev = abs(v[1] - med_pr) / (med_l2 + 1e-30)
                if v[2] > 0 or dev != dev or dev > tol:
                    print(f"step {s} run {r}: {name}  |g| {v[0]:.6e} (median {med_l2:.6e})  projection off by {dev:.3e} |g|  nan {v[2]}")
                    summary["odd"].append({"step": s, "run": r, "param": name, "dev": dev, "nan": v[2]})
    print("COLD_PROBE_SUMMARY " + json.dumps({"runs": summary["runs"], "poison": poison, "odd_entries": len(summary["odd"]),
                                              "odd_runs": sorted({o["run"] for o in summary["odd"]})}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--runs", type=int, default=12)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--tol", type=float, default=1e-5)
    a = ap.parse_args()
    if a.child:
        child(a.poison, a.steps)
    else:
        parent(a.runs, a.poison, a.steps, a.tol)
