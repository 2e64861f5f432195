"""Attention kernels at the ViT shape of the pre-training step (B=64, H=12, N=197), for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx

B, H, N, D = int(os.environ.get("B", 128)), 12, 197, 768
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda").bfloat16()
bias = torch.randn(H, N, 208, device="cuda")
dout = torch.randn(B * N, D, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
dbias = torch.zeros_like(bias)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for it in range(iters):
    o, lse = Fx.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, N, 0.125, bias=bias)
    Fx.attn_bwd(dout, qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                B, H, N, N, 0.125, bias=bias, dbias=dbias)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for it in range(iters):
    o, lse = Fx.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, N, 0.125, bias=bias)
e.record()
torch.cuda.synchronize()
print("fwd us", s.elapsed_time(e) / iters * 1e3)
s.record()
for it in range(iters):
    Fx.attn_bwd(dout, qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                B, H, N, N, 0.125, bias=bias, dbias=dbias)
e.record()
torch.cuda.synchronize()
print("bwd (dq + dkv) us", s.elapsed_time(e) / iters * 1e3)
print("checksum", float(o.float().abs().sum()), float(dqkv.float().abs().sum()), float(dbias.abs().sum()))
# single-pass delta (forward keeps o_lo)
s.record()
for it in range(iters):
    o, lse, o_lo = Fx.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, N, 0.125, bias=bias, lo=True)
e.record()
torch.cuda.synchronize()
print("fwd (+ o_lo) us", s.elapsed_time(e) / iters * 1e3)
for ph, name in ((1, "dq alone"), (2, "dkv alone"), (0, "dq + dkv")):
    for lo in (None, o_lo):
        delta = None
        if ph == 2:
            delta = Fx.attn_bwd(dout, qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                                B, H, N, N, 0.125, bias=bias, dbias=dbias, phase=1, o_lo=lo)
        s.record()
        for it in range(iters):
            Fx.attn_bwd(dout, qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                        B, H, N, N, 0.125, bias=bias, dbias=dbias, phase=ph, delta=delta, o_lo=lo)
        e.record()
        torch.cuda.synchronize()
        print(f"bwd {name:10s} {'single-pass delta' if lo is not None else 'two-pass delta  '} us", s.elapsed_time(e) / iters * 1e3)
