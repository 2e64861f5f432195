"""Which parameter gradients of the headline step are NOT bit-reproducible?

  [XFM_DETERMINISTIC=1] python tools/bit_repro.py [--reps 4] [--batch 64] [--eval-mode]

One process, the bench's model / optimizer / accelerator at the headline shape (bench.py wl_pretrain), the SAME batch every
repetition and every counter-based random draw rewound (dropout seeds, MIM mask draws, hard-negative draws, the drop-path draw on the
device generator).  Each repetition zeroes the gradient arena, runs forward + backward and snapshots the arena; repetition r is compared
with repetition 0 parameter by parameter, bit for bit.  A tensor that differs was summed in an order the hardware chose (float atomics,
a split reduction whose last writer is decided by arrival): the table names it, so the kernel can be found.  XFM_DETERMINISTIC=1 switches
the remaining float-atomic reductions to their ordered forms: the list must then be empty.  (Several STEPS, Adam included, across cold
processes: XFM_DETERMINISTIC=1 python tools/cold_probe.py --runs 5 --steps 3 --tol 1e-13.)

Prints a table and one JSON line `BIT_REPRO {...}`; exit code 1 when anything differs."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--eval-mode", action="store_true")
    ap.add_argument("--padded-rows", action="store_true")
    a = ap.parse_args()
    import bench
    from xfm_amd import xroberta as XR
    from xfm_amd.accelerators import RCCLDDPAccelerator
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    args = argparse.Namespace(batch=a.batch, pool=1, padded_rows=a.padded_rows)
    model, forward, _, _, _ = bench.wl_pretrain(args, device, 0)
    opt = bench.make_optimizer(model)
    acc = RCCLDDPAccelerator({"RNG_SEED": 42, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
    wrapped, opt, _ = acc.set_up(model, opt, None, 0, 1, 0)
    model.train(not a.eval_mode)
    arena = model._arena
    gen = model.vision_encoder.generator

    def rewind():
        XR._seed_counter[0] = 0
        gen._draws = 0
        torch.manual_seed(42)
        torch.cuda.manual_seed_all(42)

    snaps, losses = [], []
    for r in range(a.reps):
        rewind()
        total, parts = forward(wrapped, 0)
        acc.backward_step(total, opt)
        acc.grads_ready()
        torch.cuda.synchronize()
        snaps.append(arena.grad.clone())
        losses.append({k: float(v.detach()) for k, v in parts.items()})
        model.zero_grad()
        acc.grads_ready()
    what = "gradients"
    odd = {}
    for r in range(1, a.reps):
        if losses[r] != losses[0]:
            print(f"rep {r}: LOSSES differ: {losses[r]} vs {losses[0]}")
            odd.setdefault("losses", []).append(r)
        for p in arena.params:
            name = arena.names[id(p)]
            o, n = arena.offsets[id(p)]
            x, y = snaps[0][o:o + n], snaps[r][o:o + n]
            if not torch.equal(x, y):
                d = (x.double() - y.double())
                odd.setdefault(name, []).append((r, float(d.norm() / x.double().norm().clamp_min(1e-300)), int((d != 0).sum()), n))
    print(f"{what}: {len(arena.params)} tensors, {a.reps} repetitions, losses {losses[0]}")
    for name, rows in odd.items():
        if name == "losses":
            continue
        worst = max(rows, key=lambda t: t[1])
        print(f"  {name}: differs in {len(rows)} of {a.reps - 1} repetitions; worst rel-L2 {worst[1]:.2e}, {worst[2]} of {worst[3]} entries")
    import hashlib
    digest = hashlib.sha1(snaps[0].cpu().numpy().tobytes()).hexdigest()   # (across PROCESSES: compare this between runs)
    print("BIT_REPRO " + json.dumps({"what": what, "reps": a.reps, "batch": a.batch, "train_mode": not a.eval_mode, "grad_arena_sha1": digest,
                                     "deterministic_mode": os.environ.get("XFM_DETERMINISTIC", "0") not in ("", "0"),
                                     "tensors": len(arena.params), "not_bit_stable": sorted(odd)}))
    sys.exit(1 if odd else 0)


if __name__ == "__main__":
    main()
