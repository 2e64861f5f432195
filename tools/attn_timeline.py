"""Where does an entry's time go in the short attention backward's dQ kernel (ViT: 197 x 197, csrc/attention.hip
attn_bwd_dq_short_kernel)?  Wave 0 of every workgroup stamps nine points of every batch entry it walks (10-ns clock, kernel argument
`dbg` via XFM_ATTN_DBG_PTR):
  top | fetched registers + the entry's K / V DMA have landed (vmcnt 0) | barrier 1 passed | next entry's K / V LDS-DMA issued |
  S, dP, exp, delta partial done | barrier 2 passed | dS written to the exchange | barrier 3 passed | dQ summed and stored
Prints the median / p90 of every phase over all (workgroup, entry) pairs, the entry period, and the kernel time by events.
Run on the GPU box:  [B=128] python tools/attn_timeline.py"""
import os
import sys

if os.environ.get("LO", "0") == "1":
    os.environ.setdefault("XFM_ATTN_SHORT_PRE", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402


def main():
    B, H, N, D = int(os.environ.get("B", 128)), 12, 197, 768
    torch.manual_seed(0)
    qkv = torch.randn(B * N, 3 * D, device="cuda").bfloat16()
    bias = torch.randn(H, N, 208, device="cuda")
    bias_t = torch.nn.functional.pad(bias[:, :, :N].transpose(1, 2).contiguous(), (0, 208 - N)).contiguous()
    dout = torch.randn(B * N, D, device="cuda").bfloat16()
    dqkv = torch.empty_like(qkv)
    dbias = torch.zeros_like(bias)
    tiles = Fx.bias_tiles(bias, N, 0.125)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    lo = os.environ.get("LO", "0") == "1"   # keep the low half of O: the backward takes delta from dO . (O + O_lo)
    if lo:
        o, lse, o_lo = Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias, bias_tiles=tiles, lo=True)
    else:
        (o, lse), o_lo = Fx.attn_fwd(q, k, v, B, H, N, N, 0.125, bias=bias, bias_tiles=tiles), None

    def bwd():
        Fx.attn_bwd(dout, q, k, v, o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, N, N, 0.125, bias=bias, dbias=dbias,
                    bias_t=bias_t, bias_tiles=tiles, o_lo=o_lo)

    for _ in range(3):
        bwd()
    dbg = torch.zeros(256 * 32 * 16, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    wave = int(os.environ.get("WAVE", 0))   # which of the 12 waves stamps (wave = 4 * query tile + key range)
    pin = int(os.environ.get("PIN", 0))     # 1: every request of a workgroup goes to its slice's FIRST entry (cache-resident walk; garbage results)
    assert dbg.data_ptr() % 16 == 0 and 0 <= wave < 8
    os.environ["XFM_ATTN_DBG_PTR"] = hex(dbg.data_ptr() | wave | (8 if pin else 0))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    bwd()
    e.record()
    torch.cuda.synchronize()
    os.environ["XFM_ATTN_DBG_PTR"] = ""
    d = dbg.view(256, 32, 16).cpu()
    d = torch.cat([d[:, :, 0:3], d[:, :, 8:9], d[:, :, 3:8]], dim=2)   # stamp 8 (K / V prefetch issued) sits between 2 and 3
    ok = d[:, :, 8] > 0
    n_wg = int(ok[:, 0].sum())
    per_wg = ok.sum(1)[ok[:, 0]]
    print(f"B={B} LO={int(lo)} PIN={pin} stamping wave {wave}: dQ + dK/dV by events {s.elapsed_time(e) * 1e3:.1f} us; {n_wg} workgroups, {int(per_wg.min())}..{int(per_wg.max())} entries each")
    names = ["wait for the landed loads (vmcnt 0)", "barrier 1 (K / V published)", "issue of the next entry's K / V LDS-DMA", "S, dP MFMAs + exp + delta partial", "barrier 2 (delta partials)",
             "dS -> exchange", "barrier 3 (dS published)", "dQ MFMAs + store"]
    x = (d[ok] - int(d[:, 0, 0][ok[:, 0]].min())).double() / 100.0   # us since the first workgroup's start
    tot = x[:, 8] - x[:, 0]
    print(f"  entry (top -> dQ stored): median {float(tot.median()):.2f} us, p90 {float(tot.quantile(0.9)):.2f}, mean {float(tot.mean()):.2f}")
    for i, nm in enumerate(names):
        ph = x[:, i + 1] - x[:, i]
        print(f"  {nm:42s} median {float(ph.median()):.2f}  p90 {float(ph.quantile(0.9)):.2f}  mean {float(ph.mean()):.2f} us")
    # period between consecutive entries of one workgroup, and the kernel span
    per = (d[:, 1:, 0] - d[:, :-1, 0])[ok[:, 1:] & ok[:, :-1]].float() / 100.0
    t0 = int(d[:, 0, 0][ok[:, 0]].min())
    print(f"  entry period: median {float(per.median()):.2f} us, mean {float(per.mean()):.2f}; first start spread "
          f"{float((d[:, 0, 0][ok[:, 0]] - t0).max()) / 100:.1f} us; last dQ stored at {float((d[:, :, 8][ok] - t0).max()) / 100:.1f} us")
    e0 = (d[:, 0, 8] - d[:, 0, 0])[ok[:, 0]].float() / 100.0
    print(f"  first entry of a workgroup (cold K / V, bias tiles): median {float(e0.median()):.2f} us")


if __name__ == "__main__":
    main()
