"""Tile-config / split sweep for the mid-size GEMMs of the packed text / fusion towers (M ~ 1300 .. 5300 token rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


Ms = [int(x) for x in os.environ.get("MS", "5248,2624,1344").split(",")]
print("== gemm_nt (M,N,K) us per tile_hint [0 auto, 1 128x128, 2 64x128, 3 64x64, 4 ring, 5 256x256, 7 64x128x3, 8 64x64x4, 9 128x128x3, 10 128x128x4]")
for M in Ms:
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (768, 2304), (1536, 768)):
        # COLD=1: rotate through enough operand sets to defeat the 256 MiB Infinity Cache (every layer of the step brings its own)
        nbuf = max(2, int(1.2e9 / ((M * K + N * K + M * N) * 2))) if os.environ.get("COLD", "0") == "1" else 1
        As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nbuf)]
        Bs = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(nbuf)]
        Os = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nbuf)]
        res = []
        for hint in (0, 1, 2, 3, 4, 5, 7, 8, 9, 10):
            cnt = [0]

            def run():
                i = cnt[0] % nbuf
                cnt[0] += 1
                Fx.gemm_nt(As[i], Bs[i], out=Os[i], tile_hint=hint)
            us = timeit(run)
            res.append(f"{hint}:{us:6.1f}")
        del As, Bs, Os
        print(f"{M:6d} {N:5d} {K:5d} | " + " ".join(res) + f" | best {min(res, key=lambda r: float(r.split(':')[1]))}", flush=True)
if os.environ.get("NT_ONLY", "0") == "1":
    sys.exit(0)
print("== gemm_tn (M,N,K) us: auto (ring) | register-staged 128 kernel (-5) | 256-kernel forced (-3) | ring with 2 / 4 / 8 splits")
for M in Ms:
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (1536, 768)):
        dy = torch.randn(M, N, device="cuda").bfloat16()
        x = torch.randn(M, K, device="cuda").bfloat16()
        dw = torch.zeros(N, K, device="cuda")
        res = []
        for sp in (0, -5, -3, 2, 4, 8):
            try:
                us = timeit(lambda: Fx.gemm_tn(dy, x, dw, splits=sp))
                res.append(f"{sp}:{us:6.1f}")
            except Exception as ex:
                res.append(f"{sp}:  n/a ")
        print(f"{M:6d} {N:5d} {K:5d} | " + " ".join(res), flush=True)
