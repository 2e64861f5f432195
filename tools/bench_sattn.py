"""Small self-attention (text rows: B=256, H=12, S=30, key mask) -- packed kernels vs one row per workgroup (XFM_ATTN_PACK=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx

torch.manual_seed(0)
B, H, S, D = int(os.environ.get("B", 256)), 12, 30, 768
qkv = torch.randn(B * S, 3 * D, device="cuda").bfloat16()
keep = (torch.rand(B, S, device="cuda") > 0.2).to(torch.int32)
keep[:, 0] = 1
dout = torch.randn(B * S, D, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]


def fwd():
    return Fx.attn_fwd(q, k, v, B, H, S, S, 0.125, key_keep=keep)


def bwd(o, lse):
    Fx.attn_bwd(dout, q, k, v, o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, S, S, 0.125, key_keep=keep)


for _ in range(5):
    o, lse = fwd()
    bwd(o, lse)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    o, lse = fwd()
e.record()
torch.cuda.synchronize()
print("fwd us", s.elapsed_time(e) / iters * 1e3)
s.record()
for _ in range(iters):
    bwd(o, lse)
e.record()
torch.cuda.synchronize()
print("bwd (dq + dkv) us", s.elapsed_time(e) / iters * 1e3)
print("checksum", float(o.float().abs().sum()), float(dqkv.float().abs().sum()))
