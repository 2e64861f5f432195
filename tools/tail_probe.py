"""The small-tile tail of the N = 768 projections (the 3456 rows beyond the whole round of 256 x 256 tiles): the plain launch against a
K-split with fixed-order partial planes (xfm_gemm_nt_ksplit; XFM_KSPLIT_FORCE = slices, XFM_KSPLIT_TILE=1 = 128 x 128 tiles).  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402
from tools.l2_hot_probe import timeit  # noqa: E402

for M, N, K in [(3456, 768, 3072), (3456, 768, 2304), (3456, 768, 768)]:
    n = 12
    As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(n)]
    Bs = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(n)]
    bias = torch.randn(N, device="cuda")
    c = [0]

    def plain():
        i = c[0] % n
        c[0] += 1
        Fx.gemm_nt(As[i], Bs[i], bias, tile_hint=-1)

    def split():
        i = c[0] % n
        c[0] += 1
        Fx.gemm_nt_ksplit(As[i], Bs[i], bias=bias)
    tp, ts = timeit(plain, 100), timeit(split, 100)
    ref = (As[0].float() @ Bs[0].float().t() + bias)
    err = float((Fx.gemm_nt_ksplit(As[0], Bs[0], bias=bias).float() - ref).abs().max() / ref.abs().max())
    print(f"M={M} N={N} K={K}: plain {tp:.1f} us, k-split ({os.environ.get('XFM_KSPLIT_FORCE', '-')} slices, tile {os.environ.get('XFM_KSPLIT_TILE', '0')}) {ts:.1f} us, err {err:.1e}", flush=True)
