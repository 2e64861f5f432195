"""Reference point only (never on the product path): what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on the shapes where
our small-tile kernels are weakest, next to ours."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for M, N, K in [(7680, 768, 768), (7680, 3072, 768), (7680, 768, 3072), (7680, 2304, 768), (3840, 768, 768), (1920, 768, 768),
                (1920, 3072, 768), (25216, 768, 768), (25216, 3072, 768), (25216, 768, 3072), (25216, 2304, 768), (12608, 1536, 768)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda")
    ours = timeit(lambda: Fx.gemm_nt(a, b, bias))
    bt = b.t()
    lib = timeit(lambda: torch.mm(a, bt))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}   ours {ours:7.1f} us {fl / ours / 1e6:6.0f} TF   vendor {lib:7.1f} us {fl / lib / 1e6:6.0f} TF", flush=True)
