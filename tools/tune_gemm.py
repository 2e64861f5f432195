"""On-GPU sweep of the GEMM launch heuristics for the XFM shapes (tile config of gemm_nt, split count of gemm_tn)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    B = 64
    shapes_nt = [(B * 197, 2304, 768), (B * 197, 768, 768), (B * 197, 3072, 768), (B * 197, 768, 3072),
                 (B * 30, 2304, 768), (B * 30, 768, 768), (B * 30, 3072, 768), (B * 30, 768, 3072),
                 (3 * B * 30, 2304, 768), (3 * B * 30, 768, 768), (3 * B * 30, 3072, 768), (3 * B * 30, 768, 3072),
                 (3 * B * 197, 1536, 768), (B * 197, 1536, 768), (3 * B * 197, 768, 1536), (B * 15, 50265, 768), (B * 15, 768, 50304)]
    print("== gemm_nt  (M,N,K): us / TFLOP/s per tile config [auto,128x128,64x128,64x64]")
    for M, N, K in shapes_nt:
        a = torch.randn(M, K, device="cuda").bfloat16()
        b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = []
        for hint in (0, 1, 2, 3):
            us = timeit(lambda: Fx.gemm_nt(a, b, out=out, tile_hint=hint))
            res.append(f"{us:7.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF")
        print(f"{M:6d} {N:6d} {K:6d} | " + " | ".join(res), flush=True)
    print("== gemm_tn  (M,N,K): us / TFLOP/s per split count [auto,1,2,3,4,6,8,12,16]")
    shapes_tn = [(B * 197, 2304, 768), (B * 197, 768, 768), (B * 197, 3072, 768), (B * 197, 768, 3072),
                 (B * 30, 2304, 768), (B * 30, 768, 768), (B * 30, 3072, 768), (3 * B * 30, 2304, 768), (3 * B * 30, 768, 768),
                 (3 * B * 30, 3072, 768), (3 * B * 197, 1536, 768)]
    for M, N, K in shapes_tn:
        dy = torch.randn(M, N, device="cuda").bfloat16()
        x = torch.randn(M, K, device="cuda").bfloat16()
        dw = torch.zeros(N, K, device="cuda")
        res = []
        for sp in (0, 1, 2, 3, 4, 6, 8, 12, 16):
            us = timeit(lambda: Fx.gemm_tn(dy, x, dw, splits=sp))
            res.append(f"{us:6.1f}/{2.0 * M * N * K / us / 1e6:4.0f}")
        print(f"{M:6d} {N:6d} {K:6d} | " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
