"""Does the 256 x 256 NT kernel's K loop run faster when its operand panels are L2-resident?  (the K-step takes ~1.9 us
with operands that stream from the Infinity Cache; MI355X_MICROARCH.md measures L2-sourced rows at twice the Infinity-Cache rate.)
One round of 256 tiles (M = N = 4096), K = 256 / 512 (per-XCD panel footprint <= 3 MB of the 4 MB L2), timed HOT (the same operands
back to back: panels stay in every XCD's L2) and COLD-in-L2 (a ring of operand sets larger than the 32 MB of L2s, inside the 256 MB
Infinity Cache -- what a training step's GEMMs see).  The difference T(K = 512) - T(K = 256) is 4 K-steps.  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402


def timeit(fn, iters):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    M = N = int(os.environ.get("MN", "4096"))
    res = {}
    modes = tuple(os.environ.get("MODES", "hot,ic").split(","))   # (MODES=hot / MODES=ic: one operand regime per process, for the PMC passes)
    ks = tuple(int(k) for k in os.environ.get("KS", "256,512,1024,2048").split(","))
    for K in ks:
        for mode in modes:
            nbuf = 1 if mode == "hot" else max(2, int(96e6 / ((M + N) * K * 2)))
            g = torch.Generator(device="cuda").manual_seed(K)
            As = [torch.randn(M, K, device="cuda", generator=g).bfloat16() for _ in range(nbuf)]
            Bs = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(nbuf)]
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            cnt = [0]

            def run():
                i = cnt[0] % nbuf
                cnt[0] += 1
                Fx.gemm_nt(As[i], Bs[i], out=out, tile_hint=5)
            res[(K, mode)] = timeit(run, 200)
            print(f"M=N={M} K={K:5d} {mode:4s}: {res[(K, mode)]:7.2f} us  ({nbuf} operand sets)", flush=True)
    for mode in modes:
        if len(ks) < 4:
            break
        print(f"{mode}: K-step from 256->512: {(res[(512, mode)] - res[(256, mode)]) / 4:.3f} us; 512->1024: {(res[(1024, mode)] - res[(512, mode)]) / 8:.3f} us; "
              f"1024->2048: {(res[(2048, mode)] - res[(1024, mode)]) / 16:.3f} us")


if __name__ == "__main__":
    main()
