"""ViT-shape attention (B x 12 heads x 197 tokens, relative-position bias + its gradient): the batch-walking kernels of
csrc/attention_vit.hip vs the general ones (XFM_ATTN_VIT=0).  Usage: [B=128] python tools/bench_attn_vit.py [iters]"""
import os, sys
os.environ.setdefault("XFM_ATTN_VIT_BWD", "0")
if os.environ.get("LO", "0") == "1":
    os.environ.setdefault("XFM_ATTN_SHORT_PRE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx

B, H, N, D = int(os.environ.get("B", 128)), 12, 197, 768
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda").bfloat16()
bias = torch.randn(H, N, 208, device="cuda")
bias_t = bias[:, :, :N].transpose(1, 2).contiguous()
bias_t = torch.nn.functional.pad(bias_t, (0, 208 - N)).contiguous()
dout = torch.randn(B * N, D, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
dbias = torch.zeros_like(bias)
tiles = Fx.bias_tiles(bias, N, 0.125) if os.environ.get("TILED", "1") == "1" else None
if os.environ.get("NOBIAS") == "1":
    bias = bias_t = dbias = tiles = None
if os.environ.get("DBIAS", "1") == "0":   # bias, but no bias gradient
    dbias = None
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]


LO = os.environ.get("LO", "0") == "1"   # keep the low half of O: the backward takes delta from dO . (O + O_lo)
o_lo = None


def fwd():
    global o_lo
    if LO:
        o, lse, o_lo = Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias, bias_tiles=tiles, lo=True)
        return o, lse
    return Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias, bias_tiles=tiles)


def bwd(o, lse, phase=0, delta=None):
    return Fx.attn_bwd(dout, q, k, v, o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, N, N, 0.125, bias=bias, dbias=dbias, bias_t=bias_t,
                       bias_tiles=tiles, o_lo=o_lo, phase=phase, delta=delta)


for _ in range(3):
    o, lse = fwd()
    bwd(o, lse)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    o, lse = fwd()
e.record()
torch.cuda.synchronize()
tf = s.elapsed_time(e) / iters * 1e3
s.record()
for _ in range(iters):
    bwd(o, lse)
e.record()
torch.cuda.synchronize()
tb = s.elapsed_time(e) / iters * 1e3
split = ""
if os.environ.get("XFM_ATTN_VIT_BWD", "0") == "0":   # the split pair: each kernel on its own (phase 1 = dQ + delta + dbias, phase 2 = dK / dV)
    dl = bwd(o, lse, phase=1)
    ts = []
    for ph in (1, 2):
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            bwd(o, lse, phase=ph, delta=dl if ph == 2 else None)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    split = f" (dQ {ts[0]:.1f} + dK/dV {ts[1]:.1f})"
gf = 4.0 * B * H * N * N * 64 / 1e9
print(f"LO={int(LO)} XFM_ATTN_VIT={os.environ.get('XFM_ATTN_VIT', '1')} BWD={os.environ.get('XFM_ATTN_VIT_BWD')} B={B}: fwd {tf:.1f} us ({gf / tf * 1e3:.0f} TFLOP/s), bwd {tb:.1f} us{split} ({2.5 * gf / tb * 1e3:.0f} TFLOP/s); "
      f"checksum {float(o.float().abs().sum()):.4e} {float(dqkv.float().abs().sum()):.4e} {float(dbias.abs().sum()) if dbias is not None else 0.0:.4e}")
