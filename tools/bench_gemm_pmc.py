"""One GEMM shape, cold-rotated operands, for rocprofv3 --pmc runs (L2 hit rate vs tile order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfm_amd import functional as Fx

M, N, K = (int(x) for x in sys.argv[1:4])
hint = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nbuf = 8
As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nbuf)]
Bs = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(nbuf)]
Os = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nbuf)]
for i in range(16):
    Fx.gemm_nt(As[i % nbuf], Bs[i % nbuf], out=Os[i % nbuf], tile_hint=hint)
torch.cuda.synchronize()
