"""Run-to-run reproducibility of one pre-training backward (small model, eval mode): rel-L2 between the gradients of two fresh runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_util import load
from nccl_w1_worker import run

_, meta = load("pretrain_small")
a = run(meta, False)[1]
b = run(meta, False)[1]
print(f"XFM_ATTN_VIT={os.environ.get('XFM_ATTN_VIT','1')}: grads rel-L2 between two runs {float((a - b).norm() / a.norm()):.3e}, max abs {float((a-b).abs().max()):.3e}")
