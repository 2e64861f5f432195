"""Single-tile probes: per-K-tile time of the 256x256 pipelines (NT vs TN) on one CU, and the atomic epilogue's share."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for M in (6400, 25216):
    a = torch.randn(256, M, device="cuda").bfloat16()
    b = torch.randn(256, M, device="cuda").bfloat16()
    t_nt = timeit(lambda: Fx.gemm_nt(a, b, tile_hint=5))
    dy = torch.randn(M, 256, device="cuda").bfloat16()
    x = torch.randn(M, 256, device="cuda").bfloat16()
    dw = torch.zeros(256, 256, device="cuda")
    t_tn = timeit(lambda: Fx.gemm_tn(dy, x, dw, splits=-3))
    print(f"one tile, {M // 64} K-tiles: NT {t_nt:8.1f} us ({t_nt / (M // 64) * 1e3:6.0f} ns/K-tile)   TN {t_tn:8.1f} us ({t_tn / (M // 64) * 1e3:6.0f} ns/K-tile)", flush=True)
# full-chip: same work, different output sizes => different atomic volume
for (M, N, K) in ((25216, 3072, 768), (25216, 768, 768), (25216 * 4, 768, 768), (25216, 1536, 1536)):
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    for sp in (-4, -3):
        t = timeit(lambda: Fx.gemm_tn(dy, x, dw, splits=sp))
        print(f"M={M} N={N} K={K} splits={sp}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF", flush=True)
