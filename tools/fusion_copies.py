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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    bench.fusion_probe(model, 64, iters=1)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
sel = [r for r in rows if r.key in ("aten::copy_", "aten::fill_", "aten::cat", "aten::add", "aten::mul", "aten::_to_copy", "aten::zeros", "aten::add_")]
sel.sort(key=lambda r: -r.device_time_total)
for r in sel[:30]:
    print(f"{r.key:16s} calls {r.count:4d}  device {r.device_time_total:8.0f} us  shapes {r.input_shapes}")
