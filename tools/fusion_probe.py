"""The fusion encoder's forward + backward alone on the step's 4B-sequence shape (bench.fusion_probe), for rocprofv3 runs.
usage: fusion_probe.py [iters] [padded]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xfm_amd import synthetic as syn

device = torch.device("cuda", 0)
model = bench.build_model(device)
model.finalize() if hasattr(model, "finalize") else None
model.train(True)
hb = syn.pretrain_batch(64, seed=1234)
print(bench.fusion_probe(model, 64, hb, packed="padded" not in sys.argv, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 8))
