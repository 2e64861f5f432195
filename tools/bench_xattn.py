"""Grouped cross-attention (text queries over the image tokens of their key/value SOURCE) at the shapes of the bench workloads:
forward, dQ (phase 1) and dK/dV (phase 2) kernel times.  SHAPE=pretrain|retrieval|vqa python tools/bench_xattn.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xfm_amd import functional as Fx  # noqa: E402

SHAPES = {  # images, rows per image pattern, queries, keys, dropout
    "pretrain": dict(U=64, Sq=30, Sk=197, p=0.1, parts=4),     # pos | neg img | neg txt | mlm
    "retrieval": dict(U=32, Sq=40, Sk=577, p=0.1, parts=3),    # pos | neg img | neg txt
    "retrieval_pos": dict(U=32, Sq=40, Sk=577, p=0.1, parts=1),
    "vqa": dict(U=24, Sq=40, Sk=901, p=0.1, parts=1),
}
s = SHAPES[os.environ.get("SHAPE", "pretrain")]
U, Sq, Sk, p = s["U"], s["Sq"], s["Sk"], float(os.environ.get("P", s["p"]))
H, D = 12, 768
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gen = torch.Generator().manual_seed(3)
own = torch.arange(U)
idx = torch.cat([own if part != 1 else torch.randint(0, U, (U,), generator=gen) for part in range(s["parts"])]).to(torch.int32).cuda()
B = idx.numel()
q = (torch.randn(B * Sq, D, device="cuda") * 0.5).bfloat16()
kv = (torch.randn(U * Sk, 2 * D, device="cuda") * 0.5).bfloat16()
dout = (torch.randn(B * Sq, D, device="cuda") * 0.1).bfloat16()
drop = Fx.drop_params(p, 1234567)
groups = Fx.kv_groups(idx, U)
counts = torch.bincount(idx.long().cpu(), minlength=U)
dq, dkv = torch.empty_like(q), torch.empty_like(kv)
k, v, dk, dv = kv[:, :D], kv[:, D:], dkv[:, :D], dkv[:, D:]


def timed(fn, n=iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


LO = os.environ.get("LO", "1") != "0"   # the forward keeps the low half of O: one sweep over the keys in the dQ kernel (the product's form)
o, lse, *rest = Fx.attn_fwd(q, k, v, B, H, Sq, Sk, 0.125, groups=groups, drop=drop, lo=LO)
o_lo = rest[0] if LO else None
t_f = timed(lambda: Fx.attn_fwd(q, k, v, B, H, Sq, Sk, 0.125, groups=groups, drop=drop, lo=LO))
delta = Fx.attn_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, Sq, Sk, 0.125, groups=groups, drop=drop, phase=1, o_lo=o_lo)
t_q = timed(lambda: Fx.attn_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, Sq, Sk, 0.125, groups=groups, drop=drop, phase=1, o_lo=o_lo))
t_kv = timed(lambda: Fx.attn_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, Sq, Sk, 0.125, groups=groups, drop=drop, phase=2, delta=delta))
flop = 4.0 * B * Sq * Sk * 64 * H
print(f"{os.environ.get('SHAPE', 'pretrain')}: rows {B} sources {U} (rows per source: max {int(counts.max())}, mean {float(counts.float().mean()):.1f}) "
      f"Sq {Sq} Sk {Sk} p {p}:  fwd {t_f:.1f} us   dq {t_q:.1f} us   dkv {t_kv:.1f} us   ({flop / 1e9:.2f} GFLOP fwd; "
      f"K|V bytes {kv.numel() * 2 / 1e6:.1f} MB)")
print("checksum", float(o.float().abs().sum()), float(dq.float().abs().sum()), float(dkv.float().abs().sum()))
