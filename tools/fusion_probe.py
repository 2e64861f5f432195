"""The fusion encoder's forward + backward alone on the step's 4B-row shape (bench.fusion_probe), for rocprofv3 runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

device = torch.device("cuda", 0)
model = bench.build_model(device)
model.finalize() if hasattr(model, "finalize") else None
model.train(True)
print(bench.fusion_probe(model, 64, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 8))
