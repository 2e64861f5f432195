"""Host cost per kernel launch through the Python wrappers vs the bare ctypes call (tiny problems: the GPU is never the limit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx, _lib

dev = "cuda"
a = torch.randn(64, 64, device=dev).bfloat16()
b = torch.randn(64, 64, device=dev).bfloat16()
o = torch.empty(64, 64, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(64, device=dev)
lib = _lib.load()
N = 3000


def t(fn, label):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:60s} host {1e6 * (t1 - t0) / N:6.2f} us/call   (GPU drained after {1e6 * (t2 - t0) / N:6.2f} us/call)", flush=True)


st = Fx._stream()
args = (a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), o.data_ptr(), o.stride(0), bias.data_ptr(), 0, 0, 64, 64, 64, 0, 3, st)
t(lambda: lib.xfm_gemm_nt(*args), "bare ctypes xfm_gemm_nt (prebuilt args)")
t(lambda: Fx.gemm_nt(a, b, bias, out=o, tile_hint=3), "Fx.gemm_nt(out=preallocated)")
t(lambda: Fx.gemm_nt(a, b, bias, tile_hint=3), "Fx.gemm_nt (allocates its output)")
t(lambda: torch.empty((64, 64), dtype=torch.bfloat16, device=dev), "torch.empty")
t(lambda: Fx._stream(), "Fx._stream()")
w = torch.ones(768, device=dev); bb = torch.zeros(768, device=dev)
h = torch.randn(64, 768, device=dev).bfloat16(); r = torch.randn(64, 768, device=dev).bfloat16()
t(lambda: Fx.ln_post_fwd(h, r, w, bb, 1e-5), "Fx.ln_post_fwd")
t(lambda: torch.add(h, r), "torch.add (ATen elementwise, for scale)")
q = torch.randn(4 * 30, 2304, device=dev).bfloat16()
t(lambda: Fx.attn_fwd(q[:, :768], q[:, 768:1536], q[:, 1536:], 4, 12, 30, 30, 0.125), "Fx.attn_fwd")
