"""Headline benchmark: image-text pairs/s of the XFM-base pre-training step (ITC + ITM + MLM + MIM, 224 px / 30 tokens),
forward + backward (+ gradient all-reduce for N > 1, + clip/AdamW step), one process per GPU over RCCL.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     bf16-MFMA roofline of the dominant kernel (gemm_nt: forward + dgrad GEMMs), from HIP events recorded on
               the launch stream around every gemm_nt launch of one instrumented step run right after the timed region;
  cpu_baseline the CPU oracle (a port, not the reference itself) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PAIR_GFLOP = 376.1          # fwd 128.8 + bwd 247.3 GFLOP per image-text pair
BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md), never the 2:1-sparse figure

FULL_CFG = {"use_beit_v2": True, "image_res": 224, "patch_size": 16, "local_attn_depth": -1, "text_encoder": "roberta-base",
            "text_num_hidden_layers": 12, "text_fusion_start_at": 12, "fusion_num_hidden_layers": 12, "fusion_fusion_start_at": 0,
            "num_masking_patches": 75, "min_num_patches": 16, "embed_dim": 256, "temp": 0.07, "learnable_temp": True,
            "max_temp": 0.5, "min_temp": 0.001, "max_tokens": 30, "max_masks": 15}


# BASELINE.json configs -> bench workloads.  `gflop`: algorithmic fwd+bwd GFLOP per unit as the REFERENCE executes it, None
# where the survey gives no figure (only the executed count is reported then).
WORKLOADS = {
    "pretrain": dict(batch=64, cpu_batch=8, unit="pairs/s", gflop=376.1,
                     metric="image-text pairs/sec fwd+bwd, XFM-base 224px/30tok",
                     desc="Pretrain.py full multimodal step (ITC+ITM+MLM+MIM), XFM-base, synthetic 224px images + 30-token captions, random init"),
    "imagenet": dict(batch=128, cpu_batch=8, unit="images/s", gflop=105.4,
                     metric="images/sec fwd+bwd, ImageNet-1k fine-tune (ViT-only path), 224px",
                     desc="Imagenet.py:437-492 step: XFMForClassification image-only branch (BEiT-v2 base + 5-Linear head, 1000 classes), synthetic 224px images, random init"),
    "retrieval": dict(batch=32, cpu_batch=4, unit="pairs/s", gflop=580.0,
                      metric="image-text pairs/sec fwd+bwd, COCO retrieval fine-tune, 384px/40tok",
                      desc="Retrieval.py:35-74 step: XFMForRetrieval (ITC with idx soft labels + ITM with hard negatives), synthetic 384px images + 40-token captions, random init"),
    "vqa": dict(batch=24, cpu_batch=2, unit="questions/s", gflop=None,
                metric="questions/sec fwd+bwd, VQA fine-tune, 480px/40tok, k~U[1,10] answers",
                desc="VQA.py:35-72 step: XFMForVQA (ViT + text + fusion towers, 12-layer causal answer decoder, weighted answer loss), synthetic 480px images, random init"),
    "glue": dict(batch=32, cpu_batch=32, unit="sequences/s", gflop=None,
                 metric="sequences/sec fwd+bwd, GLUE MRPC fine-tune (text-only), T=128",
                 desc="run_glue.py:347-365 step: XFMForClassification text-only branch on the xbert text encoder (12 layers, 3 labels), synthetic 128-token sequences, random init"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: 2.5 s of timed steps behind 15 warm-up steps -- a fresh process on a fresh box needs ~10 steps before the caching
    # allocator has its blocks and the clocks have settled: 20 steps behind 5 read 1-1.5 ms per step slower than the steady state)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--batch", type=int, default=0, help="units per GPU (0 = the workload's configured batch; pre-train: 64, the north-star's)")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward+backward(+all-reduce) only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clocks", action="store_true", help="skip the clock / power sample under load (40 more steps after the timed region)")
    ap.add_argument("--no-fusion-probe", action="store_true", help="skip the stand-alone fusion-encoder fwd+bwd measurement (profiling runs)")
    ap.add_argument("--host-time", action="store_true", help="also print (stderr) the host's enqueue time of one step against an idle GPU")
    ap.add_argument("--cpu-batch", type=int, default=0, help="units per CPU-baseline step (0 = the workload's bounded sample; pre-train: 8, SURVEY 8d)")
    ap.add_argument("--padded-rows", action="store_true", help="push the padding rows of the 30-token captions through the text / fusion "
                    "towers like the reference does (default: unpadded token rows, xfm_amd.packing)")
    ap.add_argument("--pool", type=int, default=4, help="distinct device-resident batches rotated through the steps")
    ap.add_argument("--eval-mode", action="store_true", help="disable dropout / drop-path (not the headline setting)")
    ap.add_argument("--workload", default="pretrain", choices=sorted(WORKLOADS),
                    help="pretrain = BASELINE configs[4] shape, the headline line (default); imagenet / retrieval / vqa / glue = "
                         "configs[1] / [2] / [3] / [0] at their real shapes (secondary lines, same JSON shape)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU leg (0 = short sweep over 8/16/32/64/128)")
    a = ap.parse_args()
    if a.batch <= 0:
        a.batch = WORKLOADS[a.workload]["batch"]
    return a


def build_model(device):
    from xfm_amd.model_pretrain import XFM
    torch.manual_seed(1234)
    model = XFM(FULL_CFG)
    model.to(device)
    return model


def make_optimizer(model):
    """optim.py:4-50 via the harness mirror: AdamW (0.9, 0.98), lr 1e-4, wd 0.01, the reference's name-based no-decay rule,
    lr_mult 2 on model.init_params."""
    from xfm_amd.pretrain_loop import AttrDict, create_optimizer
    return create_optimizer(AttrDict(lr=1e-4, weight_decay=0.01, lr_mult=2), model)


class GemmTimer:
    """Wraps xfm_amd.functional.gemm_nt with HIP events on the current (= launch) stream."""

    def __init__(self):
        self.records = []

    def __enter__(self):
        from xfm_amd import functional as Fx
        import xfm_amd.beit2 as b2, xfm_amd.xroberta as xr, xfm_amd.ops as ops
        self.Fx, self.orig = Fx, Fx.gemm_nt

        import ctypes
        from xfm_amd import _lib
        lib = _lib.load()

        def plan(M, N, K, epi, hint):
            cfg, rows_a = ctypes.c_int(0), ctypes.c_int(0)
            lib.xfm_gemm_nt_plan(M, N, K, epi, hint, ctypes.byref(cfg), ctypes.byref(rows_a))
            return cfg.value, rows_a.value

        def one(a, b, bias, epi, aux, out, n, hint, cfg):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self.orig(a, b, bias, epi, aux, out, n, hint)
            e.record()
            N = b.shape[0] if n is None else n
            self.records.append((s, e, 2.0 * a.shape[0] * N * a.shape[1]))
            self.shapes.append(("nt", a.shape[0], N, a.shape[1], epi, s, e, cfg))
            return r

        def timed(a, b, bias=None, epi=0, aux=None, out=None, n=None, tile_hint=0):
            """One event pair per KERNEL: a call that the library's plan splits (whole rounds of 256x256 tiles + the remaining rows on
            the small-tile kernels, xfm_gemm_nt_plan) is issued here as exactly those two launches."""
            M, K = a.shape
            N = b.shape[0] if n is None else n
            cfg, rows_a = plan(M, N, K, epi, tile_hint)
            if rows_a == 0:
                return one(a, b, bias, epi, aux, out, n, tile_hint, cfg)
            if out is None:
                out = torch.empty((M, N), dtype=torch.float32 if epi in (1, 4) else torch.bfloat16, device=a.device)
            if epi == 2 and aux is None:
                aux = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
            one(a[:rows_a], b, bias, epi, None if aux is None else aux[:rows_a], out[:rows_a], n, 5, 5)
            cfg_b, _ = plan(M - rows_a, N, K, epi, -1)
            one(a[rows_a:], b, bias, epi, None if aux is None else aux[rows_a:], out[rows_a:], n, -1, cfg_b)
            return (out, aux) if epi == 2 else out

        def timed_tn(dy, x, dw, n=None, dbias=None, splits=0):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self.orig_tn(dy, x, dw, n=n, dbias=dbias, splits=splits)
            e.record()
            self.shapes.append(("tn", dy.shape[0], dy.shape[1] if n is None else n, x.shape[1], int(dbias is not None), s, e, 0))
            return r

        def timed_attn(kind, orig):
            def f(*args, **kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = orig(*args, **kw)
                e.record()
                off = 3 if kind == "attn_fwd" else 9  # positional (B, H, Sq, Sk)
                B_, H_, Sq_, Sk_ = args[off:off + 4]
                self.shapes.append((kind, B_ * H_, Sq_, Sk_, int(kw.get("kv_index") is not None), s, e, 0))
                return r
            return f

        def timed_tn_group(items):
            """One grouped weight-gradient launch (the deferred wgrads of a tower): one event pair, the summed 2MNK of its problems."""
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self.orig_tng(items)
            e.record()
            self.groups.append((len(items), sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in items), s, e))
            return r

        self.shapes = []
        self.groups = []
        self.orig_tn = Fx.gemm_tn
        self.orig_tng = Fx.gemm_tn_group
        self.orig_af, self.orig_ab = Fx.attn_fwd, Fx.attn_bwd
        Fx.gemm_nt = timed
        Fx.gemm_tn = timed_tn
        Fx.gemm_tn_group = timed_tn_group
        Fx.attn_fwd = timed_attn("attn_fwd", self.orig_af)
        Fx.attn_bwd = timed_attn("attn_bwd", self.orig_ab)
        return self

    def __exit__(self, *a):
        self.Fx.gemm_nt = self.orig
        self.Fx.gemm_tn = self.orig_tn
        self.Fx.gemm_tn_group = self.orig_tng
        self.Fx.attn_fwd, self.Fx.attn_bwd = self.orig_af, self.orig_ab

    def by_shape(self):
        """{(kind, M, N, K, epi): [calls, total_ms]} for the instrumented step (XFM_BENCH_GEMM_SHAPES=path dumps it)."""
        torch.cuda.synchronize()
        agg = {}
        for kind, M, N, K, epi, s, e, cfg in self.shapes:
            a = agg.setdefault((kind, M, N, K, epi), [0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e)
        return agg

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(s.elapsed_time(e) for s, e, _ in self.records)
        fl = sum(f for _, _, f in self.records)
        return len(self.records), ms, fl

    def dominant(self):
        """EVERY launch of the dominant kernel, gemm_nt_256_kernel<EPI_BF16> (configuration 5 of the library's plan with the plain bf16
        epilogue; the whole-round part of a tail-split call included): (launches, total ms, total flop, algorithmic bytes)."""
        torch.cuda.synchronize()
        n, ms, fl, by = 0, 0.0, 0.0, 0.0
        for kind, M, N, K, epi, s, e, cfg in self.shapes:
            if kind != "nt" or epi != 0 or cfg != 5:
                continue
            n += 1
            ms += s.elapsed_time(e)
            fl += 2.0 * M * N * K
            by += 2.0 * (M * K + N * K + M * N)
        return n, ms, fl, by

    def families(self):
        """{family: (launch-site calls, total ms, flop)} of the instrumented step: every kernel family the wrappers bracket."""
        torch.cuda.synchronize()
        fam = {}
        for kind, M, N, K, epi, s, e, cfg in self.shapes:
            if kind == "nt":
                name = "gemm_nt_256_kernel<EPI_BF16>" if (cfg == 5 and epi == 0) else "gemm_nt_256_kernel<GELU / dGELU epilogues>" if (cfg == 5 and epi in (2, 3)) \
                    else "gemm_nt (other tile configurations)"
                fl = 2.0 * M * N * K
            elif kind == "tn":
                name, fl = "gemm_tn family (weight gradients: 256x256 / ring kernels + their reduce)", 2.0 * M * N * K
            else:   # attn_fwd / attn_bwd: M = B * H problems of N x K scores, head_dim 64
                name = "attention forward kernels" if kind == "attn_fwd" else "attention backward kernels (dQ + dK/dV + bias gradient)"
                fl = (4.0 if kind == "attn_fwd" else 10.0) * N * K * 64 * M
            a = fam.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e)
            a[2] += fl
        for n_items, fl, s, e in self.groups:   # grouped launches: one persistent kernel (+ fix-up) for a whole tower's weight gradients
            a = fam.setdefault("gemm_tn_group_kernel (grouped weight gradients: whole 256x256 tiles + stream-K tail)", [0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e)
            a[2] += fl
        return fam


FUSION_PASS_GFLOP = 35.32    # fusion encoder fwd+bwd per (image, text) sample-pass as the reference executes it


class FlopCounter:
    """MFMA work the kernels of a code region actually execute: 2MNK per GEMM (forward / dgrad / wgrad), 4 * Sq * Sk * 64 per (batch
    row, head) of attention forward and 2.5 x that backward, with Sq = the mean REAL sequence length when the rows are packed."""

    def __enter__(self):
        from xfm_amd import functional as Fx
        self.Fx, self.flop = Fx, 0.0
        self.o = (Fx.gemm_nt, Fx.gemm_tn, Fx.attn_fwd, Fx.attn_bwd)

        def nt(a, b, bias=None, epi=0, aux=None, out=None, n=None, tile_hint=0):
            self.flop += 2.0 * a.shape[0] * (b.shape[0] if n is None else n) * a.shape[1]
            return self.o[0](a, b, bias, epi, aux, out, n, tile_hint)

        def tn(dy, x, dw, n=None, splits=0, dbias=None):
            self.flop += 2.0 * dy.shape[0] * (dy.shape[1] if n is None else n) * x.shape[1]
            return self.o[1](dy, x, dw, n=n, splits=splits, dbias=dbias)

        def att(scale_, orig, off):
            def f(*args, **kw):
                B_, H_, Sq_, Sk_ = args[off:off + 4]
                rows = args[0].shape[0] / B_ if kw.get("q_pack") is not None else Sq_   # packed: mean real length (incl. slack)
                keys = rows if (kw.get("k_pack") is not None) else Sk_
                ph = kw.get("phase", 0)
                self.flop += scale_ * 4.0 * B_ * H_ * rows * keys * 64 * (0.5 if ph in (1, 2) else 1.0)
                return orig(*args, **kw)
            return f

        def tng(items):
            self.flop += sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in items)
            return self.o_tng(items)

        self.o_tng = Fx.gemm_tn_group
        Fx.gemm_nt, Fx.gemm_tn, Fx.gemm_tn_group = nt, tn, tng
        Fx.attn_fwd, Fx.attn_bwd = att(1.0, self.o[2], 3), att(2.5, self.o[3], 9)
        return self

    def __exit__(self, *a):
        self.Fx.gemm_nt, self.Fx.gemm_tn, self.Fx.attn_fwd, self.Fx.attn_bwd = self.o
        self.Fx.gemm_tn_group = self.o_tng


def fusion_probe(model, B, host_batch, packed, iters=5):
    """The north-star's named sub-target: the fusion encoder's forward + backward alone, on the step's 4B-sequence shape (B positives,
    2B ITM negatives, B MLM rows; every sequence cross-attends one of the B images) with the caption lengths of a synthetic batch
    (U[8, 30]).  HIP events on the launch stream, outside the timed region; gradients land in the arena and are zeroed
    afterwards.  Two fractions of the MFMA peak are reported: by the FLOPs the REFERENCE spends on these 4B padded sample-passes
    (35.32 GFLOP each: throughput credit for work this build avoids -- K/V projected once per image, no padding rows) and by the
    FLOPs the kernels here actually execute (hardware utilisation)."""
    from xfm_amd.arena import grads_ready
    grads_ready()   # (this probe runs backward passes outside the accelerator: behind the last step's asynchronous zero_grad)
    from xfm_amd.packing import Pack
    dev = next(model.parameters()).device
    g = torch.Generator(device="cpu").manual_seed(7)
    T = host_batch["text_atts"].shape[1]
    lens_h = host_batch["text_atts"].sum(1)
    perm = torch.randperm(B, generator=g)
    img = (torch.randn(B, 197, 768, generator=g) * 0.7).to(dev, torch.bfloat16).requires_grad_(True)
    from xfm_amd.xfm import _ones_mask
    iatts = _ones_mask(img)  # the image mask of xfm.py:566-571 (all ones, tagged: the cross-attention skips it)
    ar = torch.arange(B, device=dev)
    index = torch.cat([ar, torch.randperm(B, generator=g).to(dev), ar, ar]).to(torch.int32)
    lens4 = torch.cat([lens_h, lens_h, lens_h[perm], lens_h])
    if packed:
        from xfm_amd.xfm import _PACK_SYNC
        ld = lens_h.to(dev).to(torch.int32)
        n_rows, t_max = int(lens_h.sum()), int(lens_h.max())
        ranges, out_rows = None, None
        if _PACK_SYNC:  # the step's default layout: every sequence exact (the drawn negatives are read back once per step),
            # sequences image by image so that each image's queries are contiguous rows (xfm.XFMBase._matching_and_fuse_mlm_packed)
            from xfm_amd.packing import image_major_layout
            seq_img = index.tolist()
            from xfm_amd.xfm import _XATTN_RANGES, _LAST_ROWS
            mpos = host_batch["masked_pos"]
            M = mpos.shape[1]
            sel_off = list(range(3 * B)) + [3 * B + j * M for j in range(B)]
            sel_len = [1] * (3 * B) + [M] * B
            pack, _, pos_of, meta, ranges = image_major_layout(lens4.tolist(), seq_img, B, T, dev, extra=(seq_img, sel_off, sel_len))
            index = meta[1].contiguous()
            if not _XATTN_RANGES:
                ranges = None
            if _LAST_ROWS and not _XATTN_RANGES:  # as the step does: the last layer only on the rows ITM ([CLS]) and MLM (masked positions) read
                start_of = pack.start.index_select(0, meta[0].long())
                sel_rows = torch.cat([start_of[:3 * B], (start_of[3 * B:, None] + mpos.to(dev, torch.int32)).reshape(-1)])
                out_rows = (sel_rows, meta[2].contiguous(), meta[3].contiguous(), M)
        else:           # no host sync: worst-case room for the negative-text block, offsets computed on the device
            pack = Pack.concat([(ld, n_rows, lens_h.tolist()), (ld, n_rows, lens_h.tolist()), (ld[perm.to(dev)], B * t_max, None),
                                (ld, n_rows, lens_h.tolist())], T)
        valid = (pack.gather_index(pack) >= 0).unsqueeze(1)           # slack rows of the negative-text block stay zero
        text = ((torch.randn(pack.cap, 768, generator=g) * 0.7).to(dev, torch.bfloat16) * valid).requires_grad_(True)
        kw = dict(encoder_embeds=text, attention_mask=None, pack=pack, encoder_row_ranges=ranges, output_rows=out_rows)
        rows = pack.cap
    else:
        text = (torch.randn(4 * B, T, 768, generator=g) * 0.7).to(dev, torch.bfloat16).requires_grad_(True)
        atts = (torch.arange(T)[None, :] < lens4[:, None]).long().to(dev)
        kw = dict(encoder_embeds=text, attention_mask=atts)
        rows = 4 * B * T

    def once():
        seq = model.fusion_encoder.bert(encoder_hidden_states=img, encoder_attention_mask=iatts, return_dict=True,
                                        encoder_batch_index=index, **kw).last_hidden_state
        seq.float().square().mean().backward()

    for _ in range(2):
        once()
    import xfm_amd.xroberta as xr
    native, xr._NATIVE_LAYERS = xr._NATIVE_LAYERS, False   # count kernel by kernel (the native per-layer call launches the same ones)
    with FlopCounter() as fc:
        once()
    xr._NATIVE_LAYERS = native
    executed = fc.flop
    best = None
    for _ in range(3):  # best of three batches of `iters` back-to-back passes (the stand-alone probe follows other work on the GPU)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            once()
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / iters
        best = t if best is None else min(best, t)
    model.zero_grad()
    ms = best
    tf_ref = 4 * B * FUSION_PASS_GFLOP / ms
    tf_exe = executed / (ms * 1e-3) / 1e12
    return {"ms": round(ms, 3), "sample_passes": 4 * B, "token_rows": rows, "token_rows_padded": 4 * B * T, "packed_rows": bool(packed),
            "last_layer_rows": (int(kw["output_rows"][0].numel()) if packed and kw.get("output_rows") is not None else rows),
            "achieved_reference_flops": round(tf_ref, 1), "frac_reference_flops": round(tf_ref / BF16_DENSE_PEAK_TFLOPS, 4),
            "achieved_executed_flops": round(tf_exe, 1), "frac_executed_flops": round(tf_exe / BF16_DENSE_PEAK_TFLOPS, 4),
            "unit": "TFLOP/s", "executed_gflop": round(executed / 1e9, 1), "reference_gflop": round(4 * B * FUSION_PASS_GFLOP, 1),
            "flop_accounting": f"reference: 4B = {4 * B} padded sample-passes x {FUSION_PASS_GFLOP} GFLOP fwd+bwd (SURVEY 8d), which re-projects "
                               "the image K/V for every sequence and pushes the padding rows through every layer; executed: 2MNK over every "
                               "GEMM launched + attention, K/V projected once per image" + (", unpadded token rows" if packed else "")}


def _oracle_params(model):
    P = {k: (v.detach().float() if v.dtype.is_floating_point else v.detach()).cpu().clone() for k, v in model.state_dict().items()}
    for k in list(P):
        if k.endswith("decoder.bias"):
            P[k] = P[k[:-len("decoder.bias")] + "bias"]
    for v in P.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    return P


def cpu_baseline(model, wl, batch_size, threads=0):
    """The CPU oracle (a port, not the reference itself) on the same architecture and synthetic batch generator; bounded sample.
    The thread count is chosen by a short sweep on a quarter-size sample (the GPU box shows 128 hardware threads but a job's CPU
    share can be far smaller: over-subscription halves the rate) and stated in `cores`."""
    from oracle import xfm_oracle as O
    P = _oracle_params(model)
    one = wl["cpu_step"]

    def timed(bs, n):
        t0 = time.time()
        for _ in range(n):
            one(O, P, bs)
            for v in P.values():
                v.grad = None
        return (time.time() - t0) / n

    ncpu = os.cpu_count() or 8
    sweep = {}
    if threads <= 0:
        cands = [c for c in (8, 16, 32, 64, 128) if c <= ncpu] or [ncpu]
        small = max(batch_size // 4, 1)
        torch.set_num_threads(cands[0])
        timed(small, 1)  # warm-up (allocator, first-touch)
        for c in cands:
            torch.set_num_threads(c)
            sweep[c] = round(small / timed(small, 1), 4)
        threads = max(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    timed(batch_size, 1)
    best = min(timed(batch_size, 1) for _ in range(2))
    gf = wl.get("gflop")
    return {"value": round(batch_size / best, 4), "unit": wl["unit"], "cores": threads, "kind": "port",
            "host_hw_threads": ncpu, "thread_sweep_quarter_sample": sweep,
            "sample": f"oracle fp32 {wl['name']} step fwd+bwd, B={batch_size}, 1 warm-up + 2 timed steps, best of 2 ({best:.2f} s/step"
                      + (f", {batch_size * gf / best / 1000:.3f} TFLOP/s)" if gf else ")"),
            "reference_cross_check": "the reference itself (unmodified models/, fp32, 8 vCPU Xeon 2.1 GHz build container): 1.02 pairs/s on the "
                                     "pre-train step at B=8 (BASELINE.md section 2)" if wl["name"] == "pretrain" else None}


# ---------------------------------------------------------------------------------------------------------------------------
# workloads: each returns (model, forward(batch_index) -> scalar loss, cpu_step(O, P, bs), extra config keys)
# ---------------------------------------------------------------------------------------------------------------------------
def _vision_ckpt_config(image_res):
    """XFMForClassification loads its vision tower from a checkpoint file (xfm.py:230-232): write a random-init one."""
    import tempfile
    from xfm_amd.beit2 import VisionTransformer
    d = tempfile.mkdtemp(prefix="xfm_bench_")
    torch.manual_seed(1234)
    v = VisionTransformer(img_size=image_res, depth=12, drop_path_rate=0.1)
    sd = dict(v.state_dict())
    sd["head.weight"], sd["head.bias"] = torch.zeros(1000, 768), torch.zeros(1000)
    torch.save({"model": sd}, os.path.join(d, "beit.pth"))
    with open(os.path.join(d, "config_beit2_base.json"), "w") as f:
        json.dump({"ckpt": os.path.join(d, "beit.pth"), "vision_width": 768, "patch_size": 16}, f)
    return os.path.join(d, "config_beit2_base.json")


def _ft_cfg(image_res, **kw):
    c = dict(FULL_CFG, image_res=image_res)
    c.update(kw)
    return c


def wl_pretrain(args, device, rank):
    from xfm_amd import synthetic as syn
    model = build_model(device)
    B = args.batch
    host = [syn.pretrain_batch(B, seed=1234 + rank + 7919 * j) for j in range(args.pool)]
    batches = [{k: v.to(device) for k, v in hb.items()} for hb in host]
    # caption lengths are host-side facts of a batch (the data loader's collate knows them): the towers run on unpadded token rows
    lens = [None if args.padded_rows else hb["text_atts"].sum(1) for hb in host]

    def forward(wrapped, j):
        batch = batches[j]
        losses = wrapped(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                         masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"], ret_mim_loss=True,
                         data_source="image", text_lens=lens[j])
        return losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"], \
            {k: losses[k] for k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")}

    def cpu_step(O, P, bs):
        b = syn.pretrain_batch(bs, seed=1234)
        out = O.pretrain_forward(P, O.default_cfg(12, 12, 12), b, ids_mask=syn.mim_block_mask(bs, 14, 75, seed=1234))
        (out["loss_itc"] + out["loss_itm"] + out["loss_mlm"] + out["loss_mim"]).backward()

    return model, forward, cpu_step, {"image_res": 224, "max_tokens": 30}, host


def wl_imagenet(args, device, rank):
    from xfm_amd import synthetic as syn
    from xfm_amd.model_classification import XFMForClassification
    torch.manual_seed(1234)
    model = XFMForClassification(_ft_cfg(224, vision_config=_vision_ckpt_config(224), task_name="imagenet", num_labels=1000,
                                         text_num_hidden_layers=0, text_fusion_start_at=0, fusion_num_hidden_layers=0)).to(device)
    B = args.batch
    g = torch.Generator().manual_seed(5 + rank)
    pool = [(syn.gaussian(f"imagenet{rank}.{j}", (B, 3, 224, 224)).to(device), torch.randint(0, 1000, (B,), generator=g).to(device))
            for j in range(args.pool)]

    def forward(wrapped, j):
        loss = wrapped(pool[j][0], None, None, pool[j][1], train=True)
        return loss, {"loss": loss}

    def cpu_step(O, P, bs):
        image = syn.gaussian("imagenet.cpu", (bs, 3, 224, 224))
        logits = O.classification_forward(P, O.default_cfg(vit_depth=12), image, None, None, deep_head=True)
        torch.nn.functional.cross_entropy(logits, torch.arange(bs) % 1000).backward()

    return model, forward, cpu_step, {"image_res": 224, "num_labels": 1000}, None


def wl_retrieval(args, device, rank):
    from xfm_amd import synthetic as syn
    from xfm_amd.model_retrieval import XFMForRetrieval
    torch.manual_seed(1234)
    model = XFMForRetrieval(_ft_cfg(384)).to(device)
    B = args.batch
    pool = []
    for j in range(args.pool):
        b = syn.pretrain_batch(B, seed=384 + rank + 7919 * j, image_res=384, max_tokens=40)
        pool.append(({k: v.to(device) for k, v in b.items()}, (torch.arange(B) + 1000 * j).to(device)))

    def forward(wrapped, j):
        b, idx = pool[j]
        itc, itm = wrapped(b["image"], b["text_ids"], b["text_atts"], idx=idx)
        return itc + itm, {"loss_itc": itc, "loss_itm": itm}

    def cpu_step(O, P, bs):
        b = syn.pretrain_batch(bs, seed=384, image_res=384, max_tokens=40)
        itc, itm = O.retrieval_forward(P, O.default_cfg(12, 12, 12), b, torch.arange(bs), [(i + 1) % bs for i in range(bs)],
                                       [(i + 2) % bs for i in range(bs)])
        (itc + itm).backward()

    return model, forward, cpu_step, {"image_res": 384, "max_tokens": 40}, None


def wl_vqa(args, device, rank):
    from types import SimpleNamespace as NS
    from xfm_amd import synthetic as syn
    from xfm_amd.model_generation import XFMForVQA
    torch.manual_seed(1234)
    model = XFMForVQA(dict(_ft_cfg(480), pad_token_id=1, decoder_fusion_start_at=0, num_dec_layers=12)).to(device)
    B = args.batch
    pool = []
    for j in range(args.pool):
        x = syn.vqa_batch(B, seed=480 + rank + 7919 * j, image_res=480)
        pool.append((x.image.to(device), NS(input_ids=x.q_ids.to(device), attention_mask=x.q_atts.to(device)),
                     NS(input_ids=x.a_ids.to(device), attention_mask=x.a_atts.to(device)), x.k, x.weights.to(device)))

    def forward(wrapped, j):
        image, q, a, k, w = pool[j]
        loss = wrapped(image, q, a, k=k, weights=w, train=True)
        return loss, {"loss": loss}

    def cpu_step(O, P, bs):
        x = syn.vqa_batch(bs, seed=480, image_res=480)
        cfg = dict(O.default_cfg(12, 12, 12), dec_layers=12, dec_fusion_start=0)
        O.vqa_train_loss(P, cfg, x.image, x.q_ids, x.q_atts, x.a_ids, x.a_atts, x.k, x.weights, 1).backward()

    return model, forward, cpu_step, {"image_res": 480, "max_tokens": 40, "answers_per_question": "U[1,10]", "answer_len": 8}, None


def wl_glue(args, device, rank):
    from xfm_amd import synthetic as syn
    from xfm_amd.model_classification import XFMForClassification
    torch.manual_seed(1234)
    model = XFMForClassification(_ft_cfg(224, text_encoder="bert-base-uncased", vision_config=_vision_ckpt_config(224), task_name="mrpc",
                                         num_labels=3, fusion_num_hidden_layers=0)).to(device)
    B, T = args.batch, 128
    vocab = model.text_encoder.config.vocab_size

    def batch(seed, bs):
        b = syn.pretrain_batch(bs, seed=seed, max_tokens=T, min_len=12, vocab=vocab, with_image=False)
        ids = b["text_ids"].clone()
        ids[b["text_atts"] == 0] = 0
        return ids, b["text_atts"], torch.arange(bs) % 3

    pool = [tuple(t.to(device) for t in batch(128 + rank + 7919 * j, B)) for j in range(args.pool)]

    def forward(wrapped, j):
        ids, atts, y = pool[j]
        loss = wrapped(None, ids, atts, y, train=True)
        return loss, {"loss": loss}

    def cpu_step(O, P, bs):
        ids, atts, y = batch(128, bs)
        feat = O.bert_model(P, "text_encoder.", ids, atts, num_layers=12, fusion_layer=12)[:, 0, :]
        torch.nn.functional.cross_entropy(O.build_mlp_forward(P, "cls_head.", feat), y).backward()

    return model, forward, cpu_step, {"max_length": 128, "num_labels": 3, "text_encoder": "xbert (bert-base-uncased shape)"}, None


BUILDERS = {"pretrain": wl_pretrain, "imagenet": wl_imagenet, "retrieval": wl_retrieval, "vqa": wl_vqa, "glue": wl_glue}


def kernel_tree_hash():
    """sha1 over the kernel sources (xfm_amd/csrc/* + the public header): profiles written by tools/hbm_traffic.py carry it, so a
    committed PMC profile that no longer describes the kernels in the tree says so in the bench line instead of going stale silently."""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "xfm_amd", "csrc")
    for f in sorted(os.listdir(csrc)) + [os.path.join("..", "..", "include", "xfm_hip.h")]:
        path = os.path.join(csrc, f)
        if os.path.isfile(path) and f.endswith((".hip", ".h")):
            with open(path, "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def sample_clocks(run_step, n_steps=40):
    """Shader / memory clock and socket power UNDER the step's load, outside the timed region: rocm-smi is read by a helper thread
    while `n_steps` more steps run (a box that measures 8 % slower than its neighbours can then be explained from its own record).
    Returns a dict of whatever rocm-smi reports for device 0 with 'clk' / 'power' / 'use' / 'temp' in its name, or a note."""
    import shutil
    import subprocess
    import threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {"note": "rocm-smi not found"}
    res = {}

    def read():
        time.sleep(0.25)   # let the queue fill
        try:
            r = subprocess.run([exe, "--showclocks", "--showpower", "--showuse", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
            d = json.loads(r.stdout)
            card = d.get("card0") or next(iter(d.values()))
            res.update({k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "socclk", "power", "gpu use", "junction", "edge"))})
        except Exception as e:   # noqa: BLE001 -- a diagnostic field must never fail the bench line
            res["note"] = f"rocm-smi failed: {type(e).__name__}: {e}"

    th = threading.Thread(target=read)
    th.start()
    t0 = time.perf_counter()
    done = 0
    while th.is_alive() or done < n_steps:
        run_step()
        done += 1
        if done % 8 == 0:
            torch.cuda.synchronize()   # keep the host within a few steps of the GPU: the sample must fall inside the busy period
        if time.perf_counter() - t0 > 30:
            break
    th.join()
    torch.cuda.synchronize()
    res["sampled_under_load_steps"] = done
    return res


def device_identity(device):
    p = torch.cuda.get_device_properties(device)
    for attr in ("uuid",):
        if hasattr(p, attr):
            return str(getattr(p, attr))
    return f"{getattr(p, 'pci_domain_id', 0)}:{getattr(p, 'pci_bus_id', -1)}:{getattr(p, 'pci_device_id', -1)}:{device.index}"


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        # invoked bare (`python3 bench.py --gpus N`): start the N ranks ourselves, as the reference's launcher does (run.py:44-75).
        # This parent has made no GPU call (importing torch does not initialise HIP); the ranks are CHILD processes of
        # torch.distributed.run and their exit code is ours.  Rank 0's JSON line passes through on stdout.
        from xfm_amd.launch import launch
        sys.exit(launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # Rehearsal knobs (not used by the driver): XFM_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and XFM_BENCH_BACKEND=gloo swaps RCCL
    # for gloo, so the N > 1 code path (arena all-reduce on the side stream, ITC all-gather, per-rank batches) can be exercised with
    # real kernels on a one-GPU box -- RCCL itself refuses two ranks on one device.  Numbers from such a run are meaningless.
    if os.environ.get("XFM_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend = os.environ.get("XFM_BENCH_BACKEND", "nccl")
    # XFM_DDP_FORCE=1 at N = 1 (no multi-GPU node to hand): a process group of ONE rank on RCCL, and the accelerator's N > 1 code path
    # runs for real -- live-set agreement, arena all-reduces launched from the tower hooks and the ViT chunk hand-over on the
    # communication stream, one grouped weight-gradient launch per handed-over chunk, the ITC all-gather.  The line's `distributed`
    # object says what RCCL saw; the value is a 1-GPU throughput WITH the data-parallel machinery's compute-side costs.
    forced = world == 1 and os.environ.get("XFM_DDP_FORCE", "0") == "1"
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29671")
        dist.init_process_group(backend, world_size=world, rank=rank)

    from xfm_amd import marks
    from xfm_amd.accelerators import RCCLDDPAccelerator

    wl = dict(WORKLOADS[args.workload], name=args.workload)
    pretrain = args.workload == "pretrain"
    model, forward, cpu_step, cfg_extra, host = BUILDERS[args.workload](args, device, rank)
    wl["cpu_step"] = cpu_step
    optimizer = make_optimizer(model)
    acc = RCCLDDPAccelerator({"RNG_SEED": 42 + rank, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
    wrapped, optimizer, _ = acc.set_up(model, optimizer, None, local_rank, world, rank)
    model.train(not args.eval_mode)

    B = args.batch
    counter = [0]
    comm_events = []

    def step(timed_comm=False):
        j = counter[0] % args.pool
        counter[0] += 1
        marks.mark("step begin")
        total, parts = forward(wrapped, j)
        marks.mark("forward end")
        if timed_comm and (world > 1 or forced):  # end of backward -> gradient all-reduce complete, on the launch stream
            acc.timing = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            comm_events.append(acc.timing)
        acc.backward_step(total, optimizer)
        marks.mark("backward end")
        if args.no_optimizer:
            model.zero_grad()
        else:
            acc.optimizer_step(optimizer, model)
        marks.mark("step end")
        return parts

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # XFM_BENCH_MAIN_PRIO=1 (A/B knob): run the step's launch stream at the high HIP priority, so that the dependent chain of the
    # towers is dispatched ahead of the weight-gradient stream it shares the CUs with
    import contextlib
    main_ctx = contextlib.nullcontext()
    if os.environ.get("XFM_BENCH_MAIN_PRIO") == "1":
        hi = torch.cuda.Stream(device=device, priority=-1)
        hi.wait_stream(torch.cuda.current_stream(device))
        main_ctx = torch.cuda.stream(hi)
    with main_ctx:
        for _ in range(args.warmup):
            losses = step()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if i == max(0, args.steps - 10):
                marks.reset()   # XFM_MARKS=1: the table of the last (up to) ten steps
            losses = step(timed_comm=True)
        barrier()
        elapsed = time.perf_counter() - t0
        if args.host_time and rank == 0:   # the host's enqueue time of one step with an empty queue in front of it (never waits for a slot)
            hs = []
            for _ in range(8):
                torch.cuda.synchronize()
                h0 = time.perf_counter()
                step()
                hs.append((time.perf_counter() - h0) * 1e3)
            torch.cuda.synchronize()
            print(f"host enqueue per step ({args.workload}): min {min(hs):.1f} ms, median {sorted(hs)[4]:.1f} ms "
                  f"against {elapsed / args.steps * 1e3:.1f} ms per free-running step", file=sys.stderr)
        if marks.ON and rank == 0:
            rows, n_avg = marks.mean_table()
            print(f"step marks (mean of the last {n_avg} steps)  | stream | queued by the host at ms | reached by the GPU at ms |", file=sys.stderr)
            for lab, st, h, g in rows:
                print(f"  {lab:24s} | {st} | {h:8.3f} | {g:8.3f} |", file=sys.stderr)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    acc.timing = None
    loss_vals = {k: round(float(v.detach()), 4) for k, v in losses.items()}
    exposed = [a.elapsed_time(b) for a, b in comm_events] if comm_events else []
    ex_stats = dict(acc.stats)

    # distributed self-description: how many ranks RCCL actually connected and on how many distinct devices
    rccl = None
    if world > 1 or forced:
        ids = [None] * world
        dist.all_gather_object(ids, device_identity(device))
        rccl = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "distinct_devices": len(set(ids)),
                "exchange_dtype": acc.exchange_dtype, "exchange_bytes": ex_stats["exchange_bytes"],
                "exchange_calls": ex_stats["exchange_calls"], "overlapped_bytes": ex_stats["overlapped_bytes"],
                "exposed_comm_ms": round(sum(exposed) / max(len(exposed), 1), 3),
                "forced_group_of_one": forced, "vit_grad_chunk_blocks": int(os.environ.get("XFM_VIT_GRAD_CHUNK", "4")),
                "exposed_comm_note": "HIP events on the launch stream: end of this rank's backward -> its last gradient all-reduce has "
                                     "landed (mean over the timed steps, rank 0); the overlapped part of the exchange ran under backward"}

    # one instrumented step (outside the timed region): HIP events around every gemm_nt launch on the launch stream
    # (weight-gradient GEMMs normally run on a second stream; for this step they stay on the launch stream so that an event pair
    # brackets exactly one kernel's execution instead of a stretch of two overlapping chains); the same step counts the MFMA work the
    # launched kernels execute (2MNK per GEMM + attention)
    from xfm_amd.xroberta import _WgradStream
    import xfm_amd.model_pretrain as mp
    import xfm_amd.xroberta as xr
    _WgradStream.enabled, text_on, mp._TEXT_STREAM_ON = False, mp._TEXT_STREAM_ON, False  # (the text tower's stream as well)
    native, xr._NATIVE_LAYERS = xr._NATIVE_LAYERS, False  # kernel by kernel, so that the wrappers see every launch (same kernels)
    with FlopCounter() as fc:
        with GemmTimer() as gt:
            step()
    _WgradStream.enabled, mp._TEXT_STREAM_ON, xr._NATIVE_LAYERS = os.environ.get("XFM_WGRAD_STREAM", "1") != "0", text_on, native
    executed_flop = fc.flop
    nlaunch, gemm_ms, gemm_flop = gt.summary()
    dom_n, dom_ms, dom_fl, dom_bytes = gt.dominant()
    dom_tf = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    if os.environ.get("XFM_BENCH_GEMM_SHAPES") and rank == 0:
        with open(os.environ["XFM_BENCH_GEMM_SHAPES"], "w") as f:
            for (kind, M, N, K, epi), (calls, ms) in sorted(gt.by_shape().items(), key=lambda kv: -kv[1][1]):
                tf = "" if kind.startswith("attn") else f"{2.0 * M * N * K * calls / ms / 1e9:6.0f}TF"
                f.write(f"{kind} M={M:6d} N={N:6d} K={K:6d} epi/bias={epi} calls={calls:3d} total={ms:7.3f}ms avg={ms / calls * 1e3:7.1f}us {tf}\n")
            for n_items, fl, s_, e_ in gt.groups:
                gms = s_.elapsed_time(e_)
                f.write(f"tn_group: {n_items} weight gradients in one grouped launch  total={gms:7.3f}ms  {fl / gms / 1e9:6.0f}TF\n")
    achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    ms_per_step = elapsed / args.steps * 1e3

    traffic, traffic_note = None, None
    if pretrain:
        for name in ("round4_hbm_traffic.json", "round3_hbm_traffic.json", "round2_hbm_traffic.json", "round1_hbm_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath):
                break
        for name5 in ("round5_hbm_traffic.json",):
            if os.path.exists(os.path.join(ROOT, "profiles", name5)):
                name, tpath = name5, os.path.join(ROOT, "profiles", name5)
        if os.path.exists(tpath):  # PMC passes are separate rocprofv3 runs of this same command (profiles/README.md)
            with open(tpath) as f:
                tj = json.load(f)
            # every instantiation of the plain-bf16 256 x 256 kernel (round 3: <0, persistent?, look-ahead>; before: <0>)
            dks = [v for k, v in tj["families"].items() if k.startswith("kernel:gemm_nt_256_kernel<0")]
            dk = {"hbm_bytes_per_step_corrected": sum(v["hbm_bytes_per_step_corrected"] for v in dks),
                  "launches_per_step": sum(v["launches_per_step"] for v in dks)} if dks else None
            if dk and dk["launches_per_step"] > 0:  # the dominant kernel's own launches (its average launch, like `achieved`)
                traffic = dk["hbm_bytes_per_step_corrected"] / dk["launches_per_step"]
                traffic_note = f"bytes beyond L2 per launch of gemm_nt_256_kernel<0>, averaged over its {dk['launches_per_step']:.0f} launches per step " \
                               f"(tail-split row blocks included), from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes in profiles/{name} " \
                               "(2 x FETCH_SIZE + WRITE_SIZE, KB; tools/pmc_traffic.sh); Infinity-Cache hits are counted; algorithmic bytes " \
                               f"per launch (A + B + C once) average {dom_bytes / max(dom_n, 1):.3e}"
                prof_hash, tree_hash = tj.get("kernel_tree_hash"), kernel_tree_hash()
                traffic_note += (f"; profile taken on kernel tree {prof_hash}, this run's tree is {tree_hash}"
                                 + ("" if prof_hash == tree_hash else " -- the kernels have changed since the profile (or it predates the stamp): "
                                    "indicative only, re-run tools/pmc_traffic.sh"))
    if rank == 0:
        gf = wl["gflop"]
        step_tf_exec = executed_flop / (ms_per_step * 1e-3) / 1e12
        fams = gt.families()
        top = max(fams.items(), key=lambda kv: kv[1][1]) if fams else None
        if not pretrain and top is not None and top[0] != "gemm_nt_256_kernel<EPI_BF16>":
            # secondary workloads: the roofline is of THEIR dominant kernel family (the one with the most kernel time in the instrumented
            # step), e.g. the weight-gradient GEMMs of the VQA answer decoder, not the headline's forward GEMM
            name, (calls, fms, ffl) = top
            tf = ffl / (fms * 1e-3) / 1e12 if fms > 0 else 0.0
            roof = {"bound": "mfma", "kernel": name + " -- the family with the most kernel time in one instrumented step of this workload",
                    "achieved": round(tf, 2), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / BF16_DENSE_PEAK_TFLOPS, 4),
                    "launches": calls, "avg_launch_us": round(fms / max(calls, 1) * 1e3, 1), "flop_per_launch_avg": ffl / max(calls, 1),
                    "kernel_ms_per_step": round(fms, 3), "traffic": None,
                    "families_ms_per_step": {k: round(v[1], 3) for k, v in sorted(fams.items(), key=lambda kv: -kv[1][1])}}
        elif dom_n > 0:
            roof = {"bound": "mfma", "kernel": "gemm_nt_256_kernel<EPI_BF16> (every launch of it in one step: forward / dgrad GEMMs, whole-round parts of tail-split calls included)",
                    "achieved": round(dom_tf, 2), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(dom_tf / BF16_DENSE_PEAK_TFLOPS, 4), "launches": dom_n,
                    "avg_launch_us": round(dom_ms / max(dom_n, 1) * 1e3, 1), "flop_per_launch_avg": dom_fl / max(dom_n, 1),
                    "traffic": traffic, "traffic_note": traffic_note}
        else:  # no launch of the 256 x 256 kernel at this size: the whole forward / dgrad GEMM family stands in
            roof = {"bound": "mfma", "kernel": "gemm_nt family (every forward / dgrad GEMM of one step; the 256 x 256 kernel is not selected at these sizes)",
                    "achieved": round(achieved, 2), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / BF16_DENSE_PEAK_TFLOPS, 4), "launches": nlaunch, "traffic": None}
        # ... and the whole gemm_nt family (every forward + dgrad GEMM of the step, small tiles included)
        roof["family_gemm_nt"] = {"achieved": round(achieved, 2), "frac": round(achieved / BF16_DENSE_PEAK_TFLOPS, 4),
                                  "calls": nlaunch, "kernel_ms_per_step": round(gemm_ms, 3), "flop_per_step": gemm_flop}
        out = {
            "metric": wl["metric"],
            "value": round(B * world * args.steps / elapsed, 2),
            "unit": wl["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": dict({"workload": wl["desc"], "units_per_gpu": B, "global_batch": B * world,
                            "dropout": not args.eval_mode, "optimizer_step_in_timed_region": not args.no_optimizer,
                            "grad_allreduce": world > 1, "parallelism": f"dp{world}"}, **cfg_extra),
            # throughput credit: the FLOPs the REFERENCE spends on this step over our time -- NOT hardware utilisation
            "step_tflops_per_gpu_reference_count": round(B * gf / ms_per_step, 2) if gf else None,
            "mfma_frac_whole_step": round(B * gf / ms_per_step / BF16_DENSE_PEAK_TFLOPS, 4) if gf else None,
            # hardware utilisation: 2MNK of every GEMM launched (forward, dgrad, wgrad) + attention, over the timed step
            "step_tflops_per_gpu_executed": round(step_tf_exec, 2),
            "mfma_frac_whole_step_executed": round(step_tf_exec / BF16_DENSE_PEAK_TFLOPS, 4),
            "executed_gflop_per_step": round(executed_flop / 1e9, 1),
            "losses_last_step": loss_vals,
            "roofline": roof,
        }
        if rccl is not None:
            out["distributed"] = rccl
        if world == 1 and not forced and not args.no_clocks:
            out["clocks_under_load"] = sample_clocks(step)
        if pretrain and not args.no_fusion_probe and world == 1 and not forced:  # single-rank only: its backward would launch unmatched gradient collectives
            out["fusion_encoder_fwd_bwd"] = fusion_probe(model, B, host[0], packed=not args.padded_rows)
        if not args.no_cpu_baseline and world == 1 and not forced:  # the CPU baseline is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(model, wl, args.cpu_batch or wl["cpu_batch"], args.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        from xfm_amd.accelerators import rccl_ddp_accelerator as _A
        if _A._HOST_TIMES and rank == 0:
            ht = sorted(_A._HOST_TIMES)
            print(f"host time per all_reduce call: n {len(ht)}, median {ht[len(ht) // 2]:.3f} ms, p90 {ht[int(len(ht) * 0.9)]:.3f} ms, max {ht[-1]:.3f} ms, "
                  f"sum per step {sum(ht) / max(counter[0], 1):.2f} ms", file=sys.stderr)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
