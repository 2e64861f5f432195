"""Where do the D2D copies of the fusion encoder's fwd+bwd come from: torch.profiler with stacks, aten::copy_ / aten::to only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

device = torch.device("cuda", 0)
model = bench.build_model(device)
model.finalize()
model.train(True)
bench.fusion_probe(model, 64, iters=1)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.fusion_probe(model, 64, iters=1)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_stack_n=12)
sel = [r for r in rows if r.key in ("aten::copy_", "aten::fill_", "aten::cat", "aten::add", "aten::mul", "aten::_to_copy")]
sel.sort(key=lambda r: -r.count)
for r in sel[:25]:
    print(r.key, r.count, f"{r.device_time_total:.0f}us")
    for fr in r.stack[:12]:
        if "xfm_amd" in fr or "bench.py" in fr:
            print("     ", fr[-110:])
            break
