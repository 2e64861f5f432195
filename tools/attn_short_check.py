"""Short-sequence attention backward (ViT 197 x 197 with bias + bias gradient): the delta-from-output path (o_lo given) against the
exchange path (no o_lo) and an fp32 torch reference of the same problem.  Run on the GPU box: python tools/attn_short_check.py [B]"""
import os
import sys

os.environ.setdefault("XFM_ATTN_SHORT_PRE", "1")   # (opt-in path: see csrc/attention.hip launch_attn_bwd)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    H, N, D = 12, int(os.environ.get("N", 197)), 768
    ld = (N + 15) // 16 * 16
    torch.manual_seed(0)
    qkv = (torch.randn(B * N, 3 * D, device="cuda") * float(os.environ.get("QSCALE", 1.0))).bfloat16()
    bias = torch.randn(H, N, ld, device="cuda")
    dout = torch.randn(B * N, D, device="cuda").bfloat16()
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse, o_lo = Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias, lo=True)
    outs = {}
    for name, lo in (("exchange", None), ("from_output", o_lo)):
        dqkv = torch.zeros_like(qkv)
        dbias = torch.zeros_like(bias)
        Fx.attn_bwd(dout, q, k, v, o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, N, N, 0.125, bias=bias, dbias=dbias, o_lo=lo)
        torch.cuda.synchronize()
        outs[name] = (dqkv[:, :D].float(), dqkv[:, D:2 * D].float(), dqkv[:, 2 * D:].float(), dbias[:, :, :N].clone())
    # fp32 reference
    qf, kf, vf = (t.float().view(B, N, H, 64).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    bf = bias[:, :, :N].clone().requires_grad_(True)
    s = qf @ kf.transpose(-1, -2) * 0.125 + bf
    out = (s.softmax(-1) @ vf).transpose(1, 2).reshape(B * N, D)
    out.backward(dout.float())
    ref = tuple(t.grad.transpose(1, 2).reshape(B * N, D) for t in (qf, kf, vf)) + (bf.grad,)
    for name, got in outs.items():
        errs = [float((g - r).norm() / r.norm()) for g, r in zip(got, ref)]
        print(f"{name:12s} rel-L2 vs fp32: dq {errs[0]:.5f} dk {errs[1]:.5f} dv {errs[2]:.5f} dbias {errs[3]:.5f}  nan {any(bool(torch.isnan(g).any()) for g in got)}")
    d = [float((a - b).norm() / b.norm()) for a, b in zip(outs["from_output"], outs["exchange"])]
    print(f"from_output vs exchange: dq {d[0]:.5f} dk {d[1]:.5f} dv {d[2]:.5f} dbias {d[3]:.5f}")


if __name__ == "__main__":
    main()
