"""Small / tail GEMM shapes: 64x128 tile with 2 vs 3 LDS stages (cold operands)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx
from tune_gemm import timeit

for M, N, K in [(3456, 768, 3072), (3456, 768, 768), (3456, 768, 2304), (7680, 768, 768), (7680, 768, 3072), (7680, 2304, 768), (1856, 1536, 768),
                (12608, 768, 1536)]:
    nbuf = max(2, int(1.5e9 / ((M * K + N * K + M * N) * 2)))
    As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nbuf)]
    Bs = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(nbuf)]
    Os = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nbuf)]
    res = []
    for hint in (0, 1, 2, 7, 3, 8):
        cnt = [0]

        def run():
            i = cnt[0] % nbuf
            cnt[0] += 1
            Fx.gemm_nt(As[i], Bs[i], out=Os[i], tile_hint=hint)
        us = timeit(run, 30)
        res.append(f"{us:6.1f}")
    print(f"{M:6d} {N:5d} {K:5d} | auto {res[0]} | 128x128 {res[1]} | 64x128 {res[2]} | 64x128/3 {res[3]} | 64x64 {res[4]} | 64x64/4 {res[5]}", flush=True)
