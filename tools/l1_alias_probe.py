"""Upper bound of anything a prefetcher / better cache placement could do for the 256 x 256 NT kernel: the same launches with every row
of A (and of B) ALIASED onto one row (row stride 0), so that each K-step's 64 KB of staging loads hit the L1 / L2 on 128 bytes of
distinct data -- the direct-to-LDS path, the LDS traffic, the barriers and the MFMAs are unchanged.  T(K) at fixed (M, N): the slope is
the K-step with "free" operands, to set against tools/kstep_probe.py.  (tools/l2_hot_probe.py cannot answer this: back-to-back launches
of one problem do NOT find their operands in L2 -- a launch's 33 MB of output evicts them; `profiles/round5_l2_hot_probe.txt` shows the
same bytes beyond L2 hot and cold.)  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402
from tools.l2_hot_probe import timeit  # noqa: E402


def main():
    for M, N in [(25216, 2304), (21760, 768), (4096, 4096)]:
        for alias in (False, True):
            pts = []
            for K in (384, 768, 1536, 3072):
                if alias:
                    a = torch.randn(1, K, device="cuda").bfloat16().expand(M, K)
                    b = (torch.randn(1, K, device="cuda") * 0.05).bfloat16().expand(N, K)
                    ring = [(a, b)]
                else:
                    n = max(2, int(200e6 / ((M + N) * K * 2)))
                    ring = [(torch.randn(M, K, device="cuda").bfloat16(), (torch.randn(N, K, device="cuda") * 0.05).bfloat16()) for _ in range(n)]
                out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                c = [0]

                def run():
                    x, w = ring[c[0] % len(ring)]
                    c[0] += 1
                    Fx.gemm_nt(x, w, out=out, tile_hint=5)
                pts.append((K // 64, timeit(run, 60)))
                del ring
            n = len(pts)
            sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
            sxx, sxy = sum(p[0] * p[0] for p in pts), sum(p[0] * p[1] for p in pts)
            slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
            icpt = (sy - slope * sx) / n
            rounds = -(-(((M + 255) // 256) * ((N + 255) // 256)) // 256)
            print(f"M={M} N={N} {'ALIASED rows (operands hit L1/L2)' if alias else 'real operands (ring of buffers)   '}: "
                  + " ".join(f"K={k * 64}:{t:.1f}us" for k, t in pts) + f" | {slope / rounds:.3f} us per K-step per executed round, fixed {icpt:.1f} us", flush=True)


if __name__ == "__main__":
    main()
