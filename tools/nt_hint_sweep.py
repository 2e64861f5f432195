import sys, os
sys.path.insert(0, "/root/repo")
import torch
from xfm_amd import functional as Fx
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K) in [(5323, 768, 768), (4388, 768, 768), (5323, 768, 3072), (4388, 768, 3072), (5323, 3072, 768), (5323, 2304, 768)]:
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = []
    for hint in range(0, 9):
        try:
            for _ in range(3):
                Fx.gemm_nt(a, b, out=out, bias=bias, tile_hint=hint)
        except Exception as e:
            res.append((hint, None)); continue
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            Fx.gemm_nt(a, b, out=out, bias=bias, tile_hint=hint)
        e.record(); torch.cuda.synchronize()
        res.append((hint, s.elapsed_time(e) / 20 * 1e3))
    print(M, N, K, " ".join(f"h{h}:{t:.1f}" if t else f"h{h}:-" for h, t in res))
