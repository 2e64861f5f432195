"""Where do the first-step gradients of the forced-collectives run differ from the plain run?  (per-parameter rel-L2)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from golden_util import load
import nccl_w1_worker as W

_, meta = load("pretrain_small")
keep = {}
orig_run = W.run
def grab(tag):
    import xfm_amd.accelerators.rccl_ddp_accelerator as R
    orig = R.RCCLDDPAccelerator.backward_step
    def bs(self, loss, opt, sync=None):
        orig(self, loss, opt, sync)
        if tag not in keep:
            torch.cuda.synchronize()
            ar = self.arena
            keep[tag] = {ar.names[id(p)]: ar.grad[o:o + n].clone() for p in ar.params for (o, n) in [ar.offsets[id(p)]]}
    R.RCCLDDPAccelerator.backward_step = bs
    return orig
import xfm_amd.accelerators.rccl_ddp_accelerator as R
o = grab("A"); W.run(meta, False); R.RCCLDDPAccelerator.backward_step = o
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
dist.init_process_group("nccl", world_size=1, rank=0)
o = grab("B"); W.run(meta, True); R.RCCLDDPAccelerator.backward_step = o
rows = []
for k, a in keep["A"].items():
    b = keep["B"][k]
    na = float(a.norm())
    if na > 0:
        rows.append((float((a - b).norm()) / na, k, na))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f"{r[0]:.3e}  {r[1]}  |g|={r[2]:.3e}")
print("tensors differing > 1e-6:", sum(1 for r in rows if r[0] > 1e-6), "of", len(rows))
dist.destroy_process_group()
