"""One wgrad shape through both TN kernels (and the NT 256x256 kernel on the mirrored shape) for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx

M, N, K = (int(x) for x in sys.argv[1:4])
nbuf = 6
dys = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(nbuf)]
xs = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nbuf)]
dw = torch.zeros(N, K, device="cuda")
for sp in (-4, -3):
    for i in range(6):
        Fx.gemm_tn(dys[i % nbuf], xs[i % nbuf], dw, splits=sp)
Bs = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(nbuf)]
Os = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nbuf)]
for i in range(6):
    Fx.gemm_nt(xs[i % nbuf], Bs[i % nbuf], out=Os[i % nbuf], tile_hint=5)
torch.cuda.synchronize()
