"""Reference point only (never on the product path): what the vendor GEMM (torch's hipBLASLt / rocBLAS back end through
torch.nn.functional.linear, bias epilogue included) reaches on THIS silicon on the step's GEMM shapes, next to the hand-written
kernels -- forward / dgrad (NT) and weight-gradient (TN) shapes, operands rotated through a ring of buffers larger than the L2s so
that both sides read Infinity-Cache / HBM-sourced operands as in the step.   python tools/blas_reference.py > profiles/roundN_vendor_gemm.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def ring(shape, n, scale=1.0):
    return [(torch.randn(shape, device="cuda") * scale).bfloat16() for _ in range(n)]


print("forward / dgrad: y = x @ W^T + b  (x [M, K], W [N, K]); vendor = torch.nn.functional.linear (bias fused by hipBLASLt)")
NT = [(25216, 2304, 768), (25216, 768, 768), (25216, 3072, 768), (25216, 768, 3072), (25216, 768, 2304), (12608, 1536, 768),
      (5241, 768, 768), (5241, 2304, 768), (5241, 3072, 768), (5241, 768, 3072), (4388, 768, 768), (4388, 3072, 768), (4388, 768, 3072),
      (2517, 2304, 768), (2517, 768, 768), (2517, 3072, 768), (1296, 768, 768), (960, 50265, 768)]
for M, N, K in NT:
    n = max(2, int(150e6 / ((M + N) * K * 2)))
    As, Bs = ring((M, K), n), ring((N, K), n, 0.05)
    bias = torch.randn(N, device="cuda")
    bias16 = bias.bfloat16()
    c = [0]

    def ours():
        i = c[0] % n
        c[0] += 1
        Fx.gemm_nt(As[i], Bs[i], bias)

    def lib():
        i = c[0] % n
        c[0] += 1
        torch.nn.functional.linear(As[i], Bs[i], bias16)
    to, tl = timeit(ours), timeit(lib)
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:6d} K={K:5d}   ours {to:7.1f} us {fl / to / 1e6:6.0f} TF   vendor {tl:7.1f} us {fl / tl / 1e6:6.0f} TF   ours/vendor time {to / tl:5.2f}", flush=True)
    del As, Bs

print("weight gradient: dW += dY^T @ X  (dY [M, N], X [M, K]); vendor = torch.addmm into an fp32... bf16 matmul dY^T @ X (no accumulate)")
TN = [(25216, 2304, 768), (25216, 768, 768), (25216, 3072, 768), (25216, 768, 3072), (5241, 768, 768), (5241, 3072, 768), (5241, 768, 3072),
      (4388, 2304, 768), (2517, 3072, 768)]
for M, N, K in TN:
    n = max(2, int(150e6 / (M * (N + K) * 2)))
    dYs, Xs = ring((M, N), n), ring((M, K), n)
    dw = torch.zeros(N, K, device="cuda")
    c = [0]

    def ours():
        i = c[0] % n
        c[0] += 1
        Fx.gemm_tn(dYs[i], Xs[i], dw)

    def lib():
        i = c[0] % n
        c[0] += 1
        torch.mm(dYs[i].t(), Xs[i])
    to, tl = timeit(ours), timeit(lib)
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:6d} K={K:5d}   ours (fp32 accumulate into dW) {to:7.1f} us {fl / to / 1e6:6.0f} TF   vendor (bf16 out) {tl:7.1f} us {fl / tl / 1e6:6.0f} TF   "
          f"ours/vendor time {to / tl:5.2f}", flush=True)
    del dYs, Xs
