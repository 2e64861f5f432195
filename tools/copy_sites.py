"""Which Python lines issue the step's small copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer blits: ~170 per step, 4 us each)?
Runs a few bench steps with torch.Tensor.copy_ / .to / .cuda / torch.as_tensor / torch.tensor wrapped, and counts, per calling line
inside xfm_amd/ (or bench.py), the calls that cross host <-> device or copy device -> device.  Run on the GPU box."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

COUNTS = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        fn = fr.filename
        if ("xfm_amd" in fn or fn.endswith("bench.py")) and "copy_sites" not in fn:
            return f"{os.path.relpath(fn)}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"


def wrap(owner, name, kind):
    orig = getattr(owner, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if ON[0]:
            try:
                src = a[1] if name == "copy_" and len(a) > 1 else (a[0] if a and isinstance(a[0], torch.Tensor) else None)
                dst = out if isinstance(out, torch.Tensor) else None
                sd = getattr(src, "device", None)
                dd = getattr(dst, "device", None)
                if dd is not None and dd.type == "cuda" and (sd is None or sd.type == "cpu" or name == "copy_" or name == "clone"):
                    n = dst.numel() * dst.element_size()
                    COUNTS[(kind + (" H2D" if sd is None or sd.type == "cpu" else " D2D"), site(), "small" if n <= 65536 else "big")] += 1
            except Exception:
                pass
        return out

    setattr(owner, name, f)


for nm in ("copy_", "to", "cuda", "clone", "contiguous"):
    wrap(torch.Tensor, nm, nm)
for nm in ("tensor", "as_tensor"):
    wrap(torch, nm, "torch." + nm)


def main():
    import argparse
    import bench
    from xfm_amd.accelerators import RCCLDDPAccelerator
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    args = argparse.Namespace(batch=64, pool=2, padded_rows=False)
    model, forward, _, _, _ = bench.wl_pretrain(args, device, 0)
    opt = bench.make_optimizer(model)
    acc = RCCLDDPAccelerator({"RNG_SEED": 42, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
    wrapped, opt, _ = acc.set_up(model, opt, None, 0, 1, 0)
    model.train()
    steps = 3
    for i in range(2 + steps):
        ON[0] = i >= 2
        total, _ = forward(wrapped, i % 2)
        acc.backward_step(total, opt)
        acc.optimizer_step(opt, model)
    torch.cuda.synchronize()
    ON[0] = False
    print(f"copies per step by calling line ({steps} steps):")
    for (kind, where, size), c in COUNTS.most_common(60):
        print(f"  {c / steps:6.1f}  {kind:14s} {size:5s}  {where}")


if __name__ == "__main__":
    main()
