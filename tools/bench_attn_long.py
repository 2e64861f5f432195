"""Attention forward / backward at the long ViT shapes of the fine-tuning configurations (384 px: 577 tokens, B = 32; 480 px: 901
tokens, B = 24): per-kernel timings of the pieces xfm_attn_bwd is made of, with and without the relative-position-bias gradient and
with the one-pass delta (forward keeps the low half of O).  N=577 B=32 python tools/bench_attn_long.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xfm_amd import functional as Fx  # noqa: E402

N = int(os.environ.get("N", 901))
B = int(os.environ.get("B", 24 if N > 600 else 32))
H, D = 12, 768
LD = (N + 15) // 16 * 16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * D, device="cuda").bfloat16()
bias = torch.randn(H, N, LD, device="cuda")
NOBIAS = os.environ.get("NOBIAS", "0") != "0"   # the same kernels without a bias (what the bias traffic costs)
dout = (torch.randn(B * N, D, device="cuda") * 0.1).bfloat16()
dqkv = torch.empty_like(qkv)
bias_t = bias[:, :, :N].transpose(1, 2).contiguous() if os.environ.get("BIAS_T", "1") != "0" else None   # [H, key, query]: the dK/dV kernels' 16-byte loads
if bias_t is not None:
    bias_t = torch.nn.functional.pad(bias_t, (0, LD - N)).contiguous()
dbias = torch.zeros_like(bias)
tiles = Fx.bias_tiles(bias, N, 0.125) if os.environ.get("TILES", "1") != "0" else None   # accumulator-layout copies (1-KB wave loads)
if NOBIAS:
    bias = bias_t = tiles = None
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
flop_fwd = 4.0 * N * N * 64 * B * H


def timed(fn, n=iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


o, lse = Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias)
o2, lse2, o_lo = Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias, lo=True)
print(f"N {N} B {B}: fwd {timed(lambda: Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias)):.1f} us"
      f"   fwd + o_lo {timed(lambda: Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias, lo=True)):.1f} us   ({flop_fwd / 1e9:.1f} GFLOP)")
for with_db in ((False,) if NOBIAS else (True, False)):
    for lo in (None, o_lo):
        db = dbias if with_db else None
        t1 = timed(lambda: Fx.attn_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, N, N, 0.125, bias=bias, dbias=db, phase=1, o_lo=lo, bias_t=bias_t, bias_tiles=tiles))
        delta = Fx.attn_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, N, N, 0.125, bias=bias, dbias=db, phase=1, o_lo=lo, bias_t=bias_t, bias_tiles=tiles)
        t2 = timed(lambda: Fx.attn_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, N, N, 0.125, bias=bias, dbias=db, phase=2, delta=delta, o_lo=lo, bias_t=bias_t, bias_tiles=tiles))
        t0 = timed(lambda: Fx.attn_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, N, N, 0.125, bias=bias, dbias=db, o_lo=lo, bias_t=bias_t, bias_tiles=tiles))
        print(f"  dbias {'yes' if with_db else 'no '}  delta {'one pass (o_lo)' if lo is not None else 'two passes     '}:  dq {t1:7.1f} us   dkv {t2:7.1f} us   "
              f"both {t0:7.1f} us  = {2.5 * flop_fwd / t0 / 1e6:.0f} TFLOP/s by 10 S^2 d")
print("checksum", float(o.float().abs().sum()), float(dqkv.float().abs().sum()), float(dbias.abs().sum()))
