"""Grouped cross-attention kernels (fusion tower) as a function of the number of sequences per image: fixed per-workgroup cost vs per-row cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx
U, H, Sq, Sk, D = 64, 12, 30, 197, 768
def timeit(fn, iters=30):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for rows_per_img in (1, 2, 4, 8):
    B = U * rows_per_img
    q = torch.randn(B * Sq, D, device="cuda").bfloat16()
    kv = torch.randn(U * Sk, 2 * D, device="cuda").bfloat16()
    dout = torch.randn(B * Sq, D, device="cuda").bfloat16()
    idx = torch.arange(U, device="cuda").repeat(rows_per_img).to(torch.int32)
    groups = Fx.kv_groups(idx, U)
    for p in (0.0, 0.1):
        drop = Fx.drop_params(p, 1234)
        o, lse = Fx.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, Sq, Sk, 0.125, groups=groups, drop=drop)
        dq, dkv = torch.empty_like(q), torch.empty((U * Sk, 2 * D), dtype=torch.bfloat16, device="cuda")
        tf = timeit(lambda: Fx.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, Sq, Sk, 0.125, groups=groups, drop=drop))
        t1 = timeit(lambda: Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Sq, Sk, 0.125, groups=groups, drop=drop, phase=1))
        delta = Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Sq, Sk, 0.125, groups=groups, drop=drop, phase=1)
        t2 = timeit(lambda: Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Sq, Sk, 0.125, groups=groups, drop=drop, phase=2, delta=delta))
        print(f"rows/image {rows_per_img} dropout {p}: fwd {tf:6.1f}  dq {t1:6.1f}  dkv {t2:6.1f} us", flush=True)
