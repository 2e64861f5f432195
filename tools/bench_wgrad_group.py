"""The weight gradients of the ViT trunk's backward (12 blocks x {fc2, fc1, proj, qkv}, M = 128 images x 197 tokens): one xfm_gemm_tn per
projection as the backward reaches it (M-split planes + a reduce each) against ONE deferred grouped call.  python tools/bench_wgrad_group.py [layers]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xfm_amd import functional as Fx  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 12
M = int(os.environ.get("M", 25216))
D = 768
shapes = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]   # (N, K) of fc2, fc1, proj, qkv
torch.manual_seed(0)
dys = {N: (torch.randn(M, N, device="cuda") * 0.1).bfloat16() for N in {s[0] for s in shapes}}
xs = {K: (torch.randn(M, K, device="cuda") * 0.5).bfloat16() for K in {s[1] for s in shapes}}
items, singles = [], []
for layer in range(L):
    for N, K in shapes:
        dw = torch.zeros(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        items.append((dys[N], xs[K], dw, db))
flop = sum(2.0 * M * dy.shape[1] * x.shape[1] for dy, x, _, _ in items)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def one_by_one():
    for dy, x, dw, db in items:
        Fx.gemm_tn(dy, x, dw, dbias=db)


t1 = timed(one_by_one)
t2 = timed(lambda: Fx.gemm_tn_group(items))
print(f"{L} blocks, M {M}: one by one {t1:.3f} ms ({flop / t1 / 1e9:.0f} TFLOP/s)   grouped {t2:.3f} ms ({flop / t2 / 1e9:.0f} TFLOP/s)")
