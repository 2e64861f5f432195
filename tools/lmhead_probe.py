"""LM head + vocabulary CE (ops.lm_head_ce) against fp64 torch on random inputs at the headline shape (960 rows x 50265): where does the
decoder-weight gradient differ (scale vs direction)?  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from xfm_amd import xroberta as XR  # noqa: E402
from xfm_amd.ops import lm_head_ce  # noqa: E402


def main():
    torch.manual_seed(0)
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 960
    m = XR.RobertaForMaskedLM(XR.RobertaConfig(num_hidden_layers=1, fusion_layer=1)).cuda().finalize()
    head = m.lm_head
    with torch.no_grad():
        head.layer_norm.bias.normal_(0, 0.3)
        head.layer_norm.weight.normal_(1, 0.2)
        head.bias.normal_(0, 0.1)
    m._arena.bump() if hasattr(m._arena, "bump") else None
    x = (torch.randn(R, 768, device="cuda") * 0.8).to(torch.bfloat16).requires_grad_(True)
    labels = torch.randint(0, 50265, (R,), device="cuda")
    labels[torch.rand(R, device="cuda") < 0.6] = -100
    loss, _ = lm_head_ce(x, head, labels, "mean")
    loss.backward()
    torch.cuda.synchronize()
    # fp64 reference on the bf16-rounded operands the kernels see
    W1, b1 = head.dense.weight.detach().to(torch.bfloat16).double(), head.dense.bias.detach().double()
    Wd, bd = head.decoder.weight.detach().to(torch.bfloat16).double().requires_grad_(True), head.bias.detach().double().requires_grad_(True)
    lw, lb = head.layer_norm.weight.detach().double(), head.layer_norm.bias.detach().double()
    xr = x.detach().double().requires_grad_(True)
    h = torch.nn.functional.gelu(xr @ W1.t() + b1)
    y = torch.nn.functional.layer_norm(h, (768,), lw, lb, head.layer_norm.eps)
    ref = torch.nn.functional.cross_entropy(y @ Wd.t() + bd, labels, ignore_index=-100)
    ref.backward()
    print(f"loss {float(loss):.6f} vs {float(ref):.6f}")
    for name, g, r in (("decoder.weight", head.decoder.weight.grad, Wd.grad), ("decoder bias", head.bias.grad, bd.grad), ("dx", x.grad, xr.grad)):
        g, r = g.double().reshape(-1), r.reshape(-1)
        a = float((g @ r) / (r @ r))
        print(f"{name}: rel-L2 {float((g - r).norm() / r.norm()):.5f}  cos {float((g @ r) / (g.norm() * r.norm())):.7f}  best scale {a:.5f}")
    g, r = head.decoder.weight.grad.double(), Wd.grad
    tgt = torch.zeros(50265, dtype=torch.bool, device="cuda")
    tgt[labels[labels >= 0]] = True
    for nm, sel in (("target rows", tgt), ("other rows", ~tgt)):
        gg, rr = g[sel].reshape(-1), r[sel].reshape(-1)
        print(f"  {nm}: rel-L2 {float((gg - rr).norm() / rr.norm()):.5f}  best scale {float((gg @ rr) / (rr @ rr)):.5f}  |ref| {float(rr.norm()):.4e}")


if __name__ == "__main__":
    main()
