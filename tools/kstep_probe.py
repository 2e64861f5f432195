"""Per-K-step time and per-tile fixed cost of the 256 x 256 NT kernel at the training geometry: T(K) for K = 384 ... 3072 at fixed (M, N),
operands from a ring of buffers (Infinity-Cache-sourced, as in the step).  slope = time per K-step per ROUND of tiles; intercept = fixed
cost (launch + prologue + epilogue + round quantisation).  Run on the GPU box:  python tools/kstep_probe.py [M N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402
from tools.l2_hot_probe import timeit  # noqa: E402


def main():
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(25216, 2304), (25216, 768), (21760, 768), (25216, 3072), (4096, 4096)]
    for M, N in shapes:
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        pts = []
        for K in (384, 768, 1536, 2304, 3072):
            nbuf = max(2, int(200e6 / ((M + N) * K * 2)))
            g = torch.Generator(device="cuda").manual_seed(K)
            As = [torch.randn(M, K, device="cuda", generator=g).bfloat16() for _ in range(nbuf)]
            Bs = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(nbuf)]
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            cnt = [0]

            def run():
                i = cnt[0] % nbuf
                cnt[0] += 1
                Fx.gemm_nt(As[i], Bs[i], out=out, tile_hint=5)
            us = timeit(run, 60)
            pts.append((K // 64, us))
            print(f"  M={M} N={N} K={K:5d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF", flush=True)
            del As, Bs
        n = len(pts)
        sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
        sxx, sxy = sum(p[0] * p[0] for p in pts), sum(p[0] * p[1] for p in pts)
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
        a = (sy - b * sx) / n
        rounds = tiles / 256
        print(f"M={M} N={N}: {tiles} tiles = {rounds:.2f} rounds; T = {a:.1f} us + {b:.3f} us x K-steps -> {b / rounds:.3f} us per K-step per full round "
              f"({b / -(-tiles // 256):.3f} per executed round), fixed {a:.1f} us", flush=True)


if __name__ == "__main__":
    main()
