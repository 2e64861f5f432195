"""Grouped cross-attention kernels at the fusion tower's shape (256 text rows x 30 queries over 64 images x 197 keys)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx

B, U, H, Sq, Sk, D = 256, 64, 12, 30, 197, 768
q = torch.randn(B * Sq, D, device="cuda").bfloat16()
kv = torch.randn(U * Sk, 2 * D, device="cuda").bfloat16()
dout = torch.randn(B * Sq, D, device="cuda").bfloat16()
ar = torch.arange(U, device="cuda")
idx = torch.cat([ar, torch.randperm(U, device="cuda"), ar, ar]).to(torch.int32)
groups = Fx.kv_groups(idx, U)
keep = torch.ones(U, Sk, dtype=torch.int32, device="cuda")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, kk, p in (("mask+dropout", keep, 0.1), ("dropout only", None, 0.1), ("mask only", keep, 0.0), ("plain", None, 0.0)):
    drop = Fx.drop_params(p, 1234)
    fast = os.environ.get("FAST", "0") == "1"   # FAST=1: delta from the forward's output halves (one sweep in the dQ kernel)
    o, lse, *rest = Fx.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, Sq, Sk, 0.125, key_keep=kk, groups=groups, drop=drop, lo=fast)
    o_lo = rest[0] if fast else None
    dq, dkv = torch.empty_like(q), torch.empty((U * Sk, 2 * D), dtype=torch.bfloat16, device="cuda")
    tf = timeit(lambda: Fx.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, Sq, Sk, 0.125, key_keep=kk, groups=groups, drop=drop, lo=fast))
    tb = timeit(lambda: Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Sq, Sk, 0.125,
                                    key_keep=kk, groups=groups, drop=drop, o_lo=o_lo))
    print(f"{name:14s} fwd {tf:6.1f} us   bwd (dq + dkv) {tb:6.1f} us", flush=True)
