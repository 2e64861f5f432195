"""Where does a tile's time go in the persistent 256 x 256 NT kernel?  Wave 0 of every workgroup stamps [tile, start, K loop done,
epilogue done] (10-ns clock) per tile (GemmNT.dbg via XFM_GEMM_DBG_PTR).  Prints, per round, the spread of starts and the medians of
K-loop time, epilogue time (issue of the stores included, their acknowledgement not), and the gap to the next tile's start.
Run on the GPU box:  python tools/tile_timeline.py M N K [epi]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfm_amd import functional as Fx  # noqa: E402


def main():
    M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (25216, 2304, 768)
    epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    g = torch.Generator(device="cuda").manual_seed(1)
    sets = [(torch.randn(M, K, device="cuda", generator=g).bfloat16(), (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()) for _ in range(3)]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aux = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi in (2, 3) else None
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    for a, b in sets:   # warm
        Fx.gemm_nt(a, b, out=out, bias=bias, epi=epi, aux=aux, tile_hint=5)
    dbg = torch.zeros(256 * 8 * 4, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    os.environ["XFM_GEMM_DBG_PTR"] = hex(dbg.data_ptr())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    Fx.gemm_nt(sets[0][0], sets[0][1], out=out, bias=bias, epi=epi, aux=aux, tile_hint=5)
    e.record()
    torch.cuda.synchronize()
    os.environ["XFM_GEMM_DBG_PTR"] = ""
    d = dbg.view(256, 8, 4).cpu()
    t0 = int(d[:, 0, 1][d[:, 0, 1] > 0].min())
    print(f"M={M} N={N} K={K} epi={epi}: kernel {s.elapsed_time(e) * 1e3:.1f} us by events; {K // 64} K-steps per tile")
    last_end = None
    for r in range(8):
        row = d[:, r, :]
        ok = row[:, 3] > 0
        if r > 0 and not bool(ok.any()):
            break
        st, kd, ed = (row[ok, 1] - t0).float() / 100, (row[ok, 2] - t0).float() / 100, (row[ok, 3] - t0).float() / 100
        line = (f"round {r}: {int(ok.sum())} workgroups; start {float(st.min()):.1f} .. {float(st.median()):.1f} .. {float(st.max()):.1f} us; "
                f"K loop median {float((kd - st).median()):.2f} us ({float((kd - st).median()) / (K // 64):.3f} per K-step; max {float((kd - st).max()):.2f}); "
                f"epilogue median {float((ed - kd).median()):.2f} (max {float((ed - kd).max()):.2f}); end median {float(ed.median()):.1f} max {float(ed.max()):.1f}")
        if last_end is not None:
            both = ok & (d[:, r - 1, 3] > 0)
            gap = (d[both, r, 1] - d[both, r - 1, 3]).float() / 100
            line += f"; gap after previous epilogue median {float(gap.median()):.2f}"
        last_end = ed
        print(line)


if __name__ == "__main__":
    main()
