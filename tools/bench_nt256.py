"""256 x 256 NT kernel on the step's big shapes, cold operands: time per epilogue, and (CHECK=1) bitwise comparison of the
persistent launch (default) with the one-workgroup-per-tile launch, which a child process computes with XFM_GEMM_PERSIST=0.

    python tools/bench_nt256.py            # timings of this process's mode (XFM_GEMM_PERSIST)
    CHECK=1 python tools/bench_nt256.py    # + outputs hashed; run twice with XFM_GEMM_PERSIST=0/1 and compare the hash lines
"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402

SHAPES = [(25216, 3072, 768, 2), (25216, 3072, 768, 3), (25216, 2304, 768, 0), (25216, 768, 3072, 0), (25216, 768, 768, 0),
          (25216, 768, 2304, 0), (12608, 3072, 768, 2), (12608, 2304, 768, 0), (10752, 1536, 768, 0), (5214, 3072, 768, 2),
          (25000, 3000, 768, 2), (25000, 3000, 768, 3), (25000, 3000, 768, 0), (6000, 3072, 64, 0), (6000, 3072, 128, 2)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    check = os.environ.get("CHECK", "0") == "1"
    print("XFM_GEMM_PERSIST =", os.environ.get("XFM_GEMM_PERSIST", "(default 1)"))
    for M, N, K, epi in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(M + N + K + epi)
        nbuf = max(2, int(1.2e9 / ((M * K + N * K + 2 * M * N) * 2)))
        As = [torch.randn(M, K, device="cuda", generator=g).bfloat16() for _ in range(nbuf)]
        Bs = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(nbuf)]
        bias = torch.randn(N, device="cuda", generator=g) * 0.1
        ldc = (N + 7) // 8 * 8
        Os = [torch.empty(M, ldc, device="cuda", dtype=torch.bfloat16)[:, :N] for _ in range(nbuf)]
        aux = [(torch.randn(M, ldc, device="cuda", generator=g).bfloat16() if epi == 3 else torch.empty(M, ldc, device="cuda", dtype=torch.bfloat16))[:, :N]
               for _ in range(nbuf)]
        cnt = [0]

        def run():
            i = cnt[0] % nbuf
            cnt[0] += 1
            Fx.gemm_nt(As[i], Bs[i], out=Os[i], bias=bias, epi=epi, aux=aux[i] if epi in (2, 3) else None, tile_hint=5)
        us = timeit(run)
        line = f"{M:6d} {N:5d} {K:5d} epi {epi} | {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF"
        if check:
            cnt[0] = 0
            run()
            torch.cuda.synchronize()
            h = hashlib.sha1(Os[0].contiguous().view(torch.int16).cpu().numpy().tobytes())
            if epi == 2:
                h.update(aux[0].contiguous().view(torch.int16).cpu().numpy().tobytes())
            line += " | sha1 " + h.hexdigest()[:16]
            if epi == 0:   # against fp32 math on the first / last 300 rows (any K order must agree to bf16 rounding)
                rows = torch.cat([torch.arange(300), torch.arange(M - 300, M)]).cuda()
                ref = As[0][rows].float() @ Bs[0].float().t() + bias
                err = (Os[0][rows].float() - ref).abs().max().item() / ref.abs().max().item()
                line += f" | max err / max |ref| {err:.2e}"
        print(line, flush=True)
        del As, Bs, Os, aux


if __name__ == "__main__":
    main()
