"""The ViT trunk's grouped weight gradients (48 problems over M = 25216 rows: qkv, proj, fc1, fc2 of 12 blocks) as ONE xfm_gemm_tn_group
call, as the step launches them: time and TFLOP/s.  XFM_TN_SYNC_WINDOW=<K-steps> turns the XCD progress throttle on (csrc/gemm.hip
Tn256Seg).  Run on the GPU box: [XFM_TN_SYNC_WINDOW=4] python tools/tn_group_bench.py [blocks=12] [M=25216]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 25216
    g = torch.Generator(device="cuda").manual_seed(3)
    items, flop = [], 0
    for _ in range(blocks):
        x768 = torch.randn(M, 768, device="cuda", generator=g).bfloat16()
        x3072 = torch.randn(M, 3072, device="cuda", generator=g).bfloat16()
        for N, x in ((2304, x768), (768, x768), (3072, x768), (768, x3072)):
            dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
            dw = torch.zeros(N, x.shape[1], device="cuda")
            db = torch.zeros(N, device="cuda")
            items.append((dy, x, dw, db))
            flop += 2 * M * N * x.shape[1]
    for _ in range(2):
        Fx.gemm_tn_group(items)
    torch.cuda.synchronize()
    ref = [it[2].clone() for it in items[:4]]
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        s.record()
        Fx.gemm_tn_group(items)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    # (dw accumulates: 7 calls in all; the first four problems against a plain matmul)
    err = 0.0
    for (dy, x, dw, db), _ in zip(items[:4], ref):
        want = 7.0 * (dy.float().t() @ x.float())
        err = max(err, float((dw - want).norm() / want.norm()))
    print(f"XFM_TN_SYNC_WINDOW={os.environ.get('XFM_TN_SYNC_WINDOW', '0')}: {len(items)} problems, M = {M}: median {ts[2]:.3f} ms (min {ts[0]:.3f}) "
          f"= {flop / ts[2] / 1e9:.0f} TFLOP/s; rel error of dW vs fp32 matmul {err:.2e}")


if __name__ == "__main__":
    main()
