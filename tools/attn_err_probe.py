"""Accuracy of the ViT-shape forward attention (XFM_ATTN_VIT=1: batch-walking kernel, 0: general kernel) against fp64 math."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xfm_amd import functional as Fx
B, H, N, D = 4, 12, 197, 768
g = torch.Generator(device="cpu").manual_seed(0)
qkv = (torch.randn(B * N, 3 * D, generator=g) * float(os.environ.get("QSCALE", "1.0"))).cuda().bfloat16()
bias = torch.zeros(H, N, 208, device="cuda"); bias[:, :, :N] = torch.randn(H, N, N, generator=g).cuda()
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
tiles = Fx.bias_tiles(bias, N, 0.125) if os.environ.get("TILED", "1") == "1" else None
o, lse = Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias, bias_tiles=tiles)
qh = q.double().view(B, N, H, 64).permute(0, 2, 1, 3); kh = k.double().view(B, N, H, 64).permute(0, 2, 1, 3); vh = v.double().view(B, N, H, 64).permute(0, 2, 1, 3)
s = qh @ kh.transpose(-1, -2) * 0.125 + bias[:, :, :N].double()
ref = (s.softmax(-1) @ vh).permute(0, 2, 1, 3).reshape(B * N, D)
err = o.double() - ref
print(f"XFM_ATTN_VIT={os.environ.get('XFM_ATTN_VIT','1')} TILED={os.environ.get('TILED','1')} QSCALE={os.environ.get('QSCALE','1.0')}: out rel-L2 {float(err.norm() / ref.norm()):.3e}  max abs {float(err.abs().max()):.3e}  "
      f"lse max abs err {float((lse[:, :, :N].double() - s.logsumexp(-1)).abs().max()):.3e}  mean signed err {float(err.mean()):.3e}")
