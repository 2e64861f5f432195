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
    B = 128
    cold = len(sys.argv) > 1 and sys.argv[1] == 'cold'
    shapes_nt = [(B * 197, 2304, 768), (B * 197, 768, 768), (B * 197, 3072, 768), (B * 197, 768, 3072),
                 (B * 30, 2304, 768), (B * 30, 768, 768), (B * 30, 3072, 768), (B * 30, 768, 3072),
                 (3 * B * 30, 2304, 768), (3 * B * 30, 768, 768), (3 * B * 30, 3072, 768), (3 * B * 30, 768, 3072),
                 (3 * B * 197, 1536, 768), (B * 197, 1536, 768), (3 * B * 197, 768, 1536), (B * 15, 50265, 768), (B * 15, 768, 50304)]
    print("== gemm_nt  (M,N,K): us / TFLOP/s per tile config [auto,128x128,ring256x128,256x256]")
    if os.environ.get('XFM_TUNE_NT_SHAPES'):
        shapes_nt = [(1920, 768, 768), (1920, 2304, 768), (1920, 3072, 768), (1920, 768, 3072), (1920, 768, 2304), (7680, 768, 768), (7680, 2304, 768), (7680, 768, 3072), (960, 768, 768)]
    for M, N, K in shapes_nt:
        # cold mode: rotate through enough distinct operand sets to defeat the 256 MiB Infinity Cache (as in the real step,
        # where every layer brings its own activations and weights from HBM)
        nbuf = max(2, int(1.5e9 / ((M * K + N * K + M * N) * 2))) if cold else 1
        As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nbuf)]
        Bs = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(nbuf)]
        Os = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nbuf)]
        res = []
        for hint in ((0, 2, 3, 7, 8) if os.environ.get('XFM_TUNE_NT_SHAPES') else (0, 1, 4, 5)):
            cnt = [0]

            def run():
                i = cnt[0] % nbuf
                cnt[0] += 1
                Fx.gemm_nt(As[i], Bs[i], out=Os[i], tile_hint=hint)
            us = timeit(run)
            res.append(f"{us:7.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF")
        del As, Bs, Os
        print(f"{M:6d} {N:6d} {K:6d} | " + " | ".join(res), flush=True)
    if os.environ.get("XFM_TUNE_NT_SHAPES"):
        return
    print("== gemm_tn  (M,N,K): us / TFLOP/s [auto, 128x128 kernel forced (atomics, no workspace)]")
    shapes_tn = [(B * 197, 2304, 768), (B * 197, 768, 768), (B * 197, 3072, 768), (B * 197, 768, 3072),
                 (1920, 2304, 768), (1920, 768, 768), (1920, 3072, 768), (1920, 768, 3072), (7680, 2304, 768), (7680, 768, 768),
                 (7680, 3072, 768), (12608, 1536, 768), (960, 50304, 768)]
    for M, N, K in shapes_tn:
        nbuf = max(2, int(1.5e9 / ((M * K + M * N) * 2))) if cold else 1
        dys = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(nbuf)]
        xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nbuf)]
        dw = torch.zeros(N, K, device="cuda")
        res = []
        for sp in (0, -4):
            cnt = [0]

            def run():
                i = cnt[0] % nbuf
                cnt[0] += 1
                Fx.gemm_tn(dys[i], xs[i], dw, splits=sp)
            us = timeit(run)
            res.append(f"{us:6.1f}/{2.0 * M * N * K / us / 1e6:4.0f}")
        del dys, xs
        print(f"{M:6d} {N:6d} {K:6d} | " + " | ".join(res), flush=True)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "epi"):
    main()


def epilogues():
    """GELU / DGELU epilogue variants, cold operands: 128x128 (two workgroups per CU) vs the 256x128 ring (one per CU)."""
    M, N, K = 25216, 3072, 768
    nbuf = 6
    As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nbuf)]
    Bs = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(nbuf)]
    Os = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nbuf)]
    Us = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(nbuf)]
    bias = torch.randn(N, device="cuda")
    for name, epi in (("plain", Fx.EPI_BF16), ("gelu", Fx.EPI_GELU), ("dgelu", Fx.EPI_DGELU)):
        res = []
        for hint in (1, 4, 5):
            cnt = [0]

            def run():
                i = cnt[0] % nbuf
                cnt[0] += 1
                Fx.gemm_nt(As[i], Bs[i], bias if epi != Fx.EPI_DGELU else None, epi=epi, aux=Us[i], out=Os[i], tile_hint=hint)
            us = timeit(run)
            res.append(f"hint{hint}: {us:7.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF")
        print(f"{name:6s} {M}x{N}x{K} | " + " | ".join(res), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "epi":
    epilogues()
