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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU (north-star: 64)")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward+backward(+all-reduce) only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fusion-probe", action="store_true", help="skip the stand-alone fusion-encoder fwd+bwd measurement (profiling runs)")
    ap.add_argument("--cpu-batch", type=int, default=8, help="pairs per CPU-baseline step (SURVEY 8d: B = 8, 1 warm-up + 3 timed)")
    ap.add_argument("--padded-rows", action="store_true", help="push the padding rows of the 30-token captions through the text / fusion "
                    "towers like the reference does (default: unpadded token rows, xfm_amd.packing)")
    ap.add_argument("--pool", type=int, default=4, help="distinct device-resident batches rotated through the steps")
    ap.add_argument("--eval-mode", action="store_true", help="disable dropout / drop-path (not the headline setting)")
    return ap.parse_args()


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

        self.shapes = []
        self.orig_tn = Fx.gemm_tn
        self.orig_af, self.orig_ab = Fx.attn_fwd, Fx.attn_bwd
        Fx.gemm_nt = timed
        Fx.gemm_tn = timed_tn
        Fx.attn_fwd = timed_attn("attn_fwd", self.orig_af)
        Fx.attn_bwd = timed_attn("attn_bwd", self.orig_ab)
        return self

    def __exit__(self, *a):
        self.Fx.gemm_nt = self.orig
        self.Fx.gemm_tn = self.orig_tn
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

        Fx.gemm_nt, Fx.gemm_tn = nt, tn
        Fx.attn_fwd, Fx.attn_bwd = att(1.0, self.o[2], 3), att(2.5, self.o[3], 9)
        return self

    def __exit__(self, *a):
        self.Fx.gemm_nt, self.Fx.gemm_tn, self.Fx.attn_fwd, self.Fx.attn_bwd = self.o


def fusion_probe(model, B, host_batch, packed, iters=5):
    """The north-star's named sub-target: the fusion encoder's forward + backward alone, on the step's 4B-sequence shape (B positives,
    2B ITM negatives, B MLM rows; every sequence cross-attends one of the B images) with the caption lengths of a synthetic batch
    (U[8, 30]).  HIP events on the launch stream, outside the timed region; gradients land in the arena and are zeroed
    afterwards.  Two fractions of the MFMA peak are reported: by the FLOPs the REFERENCE spends on these 4B padded sample-passes
    (35.32 GFLOP each: throughput credit for work this build avoids -- K/V projected once per image, no padding rows) and by the
    FLOPs the kernels here actually execute (hardware utilisation)."""
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


def cpu_baseline(model, batch_size):
    """The CPU oracle on the same architecture, same synthetic batch generator; bounded sample."""
    from oracle import xfm_oracle as O
    from xfm_amd import synthetic as syn
    P = {k: (v.detach().float() if v.dtype.is_floating_point else v.detach()).cpu().clone() for k, v in model.state_dict().items()}
    for k in list(P):
        if k.endswith("decoder.bias"):
            P[k] = P[k[:-len("decoder.bias")] + "bias"]
    for v in P.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    cfg = O.default_cfg(12, 12, 12)
    b = syn.pretrain_batch(batch_size, seed=1234)
    masks = syn.mim_block_mask(batch_size, 14, 75, seed=1234)
    times = []
    for it in range(4):
        t0 = time.time()
        out = O.pretrain_forward(P, cfg, b, ids_mask=masks)
        total = out["loss_itc"] + out["loss_itm"] + out["loss_mlm"] + out["loss_mim"]
        total.backward()
        for v in P.values():
            v.grad = None
        times.append(time.time() - t0)
    best = min(times[1:])
    return {"value": round(batch_size / best, 4), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32 full pre-train step fwd+bwd, B={batch_size}, 1 warm-up + 3 timed steps, best of 3 "
                      f"({best:.2f} s/step, {batch_size * PAIR_GFLOP / best / 1000:.3f} TFLOP/s)"}


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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("XFM_BENCH_BACKEND", "nccl"), world_size=world, rank=rank)

    from xfm_amd import synthetic as syn
    from xfm_amd.accelerators import RCCLDDPAccelerator

    model = build_model(device)
    optimizer = make_optimizer(model)
    acc = RCCLDDPAccelerator({"RNG_SEED": 42 + rank, "CLIP_GRAD_NORM": 1.0, "GRAD_ACCUMULATE_STEPS": 1})
    wrapped, optimizer, _ = acc.set_up(model, optimizer, None, local_rank, world, rank)
    model.train(not args.eval_mode)

    B = args.batch
    # a small pool of distinct device-resident batches, rotated step by step (inputs are in HBM before the timed region; one batch
    # replayed for the whole run would be trained to convergence on, e.g. an ITC loss of 0.01 after 25 steps)
    host = [syn.pretrain_batch(B, seed=1234 + rank + 7919 * j) for j in range(args.pool)]
    batches = [{k: v.to(device) for k, v in hb.items()} for hb in host]
    # caption lengths are host-side facts of a batch (the data loader's collate knows them): the towers run on unpadded token rows
    lens = [None if args.padded_rows else hb["text_atts"].sum(1) for hb in host]
    counter = [0]

    def step():
        j = counter[0] % len(batches)
        batch = batches[j]
        counter[0] += 1
        losses = wrapped(batch["image"], batch["text_ids"], batch["text_atts"], text_ids_masked=batch["text_ids_masked"],
                         masked_pos=batch["masked_pos"], masked_ids=batch["masked_ids"], ret_mim_loss=True,
                         data_source="image", text_lens=lens[j])
        total = losses["loss_itc"] + losses["loss_itm"] + losses["loss_mlm"] + losses["loss_mim"]
        acc.backward_step(total, optimizer)
        if args.no_optimizer:
            model.zero_grad()
        else:
            acc.optimizer_step(optimizer, model)
        return losses

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        losses = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    loss_vals = {k: round(float(v.detach()), 4) for k, v in losses.items() if k in ("loss_itc", "loss_itm", "loss_mlm", "loss_mim")}

    # one instrumented step (outside the timed region): HIP events around every gemm_nt launch on the launch stream
    # (weight-gradient GEMMs normally run on a second stream; for this step they stay on the launch stream so that an event pair
    # brackets exactly one kernel's execution instead of a stretch of two overlapping chains)
    from xfm_amd.xroberta import _WgradStream
    import xfm_amd.model_pretrain as mp
    import xfm_amd.xroberta as xr
    _WgradStream.enabled, text_on, mp._TEXT_STREAM_ON = False, mp._TEXT_STREAM_ON, False  # (the text tower's stream as well)
    native, xr._NATIVE_LAYERS = xr._NATIVE_LAYERS, False  # kernel by kernel, so that the wrappers see every launch (same kernels)
    with GemmTimer() as gt:
        step()
    _WgradStream.enabled, mp._TEXT_STREAM_ON, xr._NATIVE_LAYERS = os.environ.get("XFM_WGRAD_STREAM", "1") != "0", text_on, native
    nlaunch, gemm_ms, gemm_flop = gt.summary()
    dom_n, dom_ms, dom_fl, dom_bytes = gt.dominant()
    dom_tf = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    if os.environ.get("XFM_BENCH_GEMM_SHAPES") and rank == 0:
        with open(os.environ["XFM_BENCH_GEMM_SHAPES"], "w") as f:
            for (kind, M, N, K, epi), (calls, ms) in sorted(gt.by_shape().items(), key=lambda kv: -kv[1][1]):
                tf = "" if kind.startswith("attn") else f"{2.0 * M * N * K * calls / ms / 1e9:6.0f}TF"
                f.write(f"{kind} M={M:6d} N={N:6d} K={K:6d} epi/bias={epi} calls={calls:3d} total={ms:7.3f}ms avg={ms / calls * 1e3:7.1f}us {tf}\n")
    achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    ms_per_step = elapsed / args.steps * 1e3

    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "round2_hbm_traffic.json")
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", "round1_hbm_traffic.json")
    if os.path.exists(tpath):  # PMC passes are separate rocprofv3 runs of this same command (profiles/README.md)
        with open(tpath) as f:
            tj = json.load(f)
        dk = tj["families"].get("kernel:gemm_nt_256_kernel<0>")
        fam = tj["families"].get("gemm_nt")
        if dk and dk["launches_per_step"] > 0:  # the dominant kernel's own launches (its average launch, like `achieved`)
            traffic = dk["hbm_bytes_per_step_corrected"] / dk["launches_per_step"]
            traffic_note = f"bytes beyond L2 per launch of gemm_nt_256_kernel<0>, averaged over its {dk['launches_per_step']:.0f} launches per step " \
                           "(tail-split row blocks included), from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes " \
                           "(2 x FETCH_SIZE + WRITE_SIZE, KB; tools/pmc_traffic.sh); Infinity-Cache hits are counted; algorithmic bytes " \
                           f"per launch (A + B + C once) average {dom_bytes / max(dom_n, 1):.3e}"
        elif fam and fam["launches_per_step"] >= nlaunch:  # kernel launches >= host calls: a call may split into two launches
            traffic = fam["hbm_bytes_per_step_corrected"]
            traffic_note = f"bytes beyond L2 per STEP over all gemm_nt kernels ({fam['launches_per_step']:.0f} launches for the family's {nlaunch} calls), " \
                           "from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (2 x FETCH_SIZE + WRITE_SIZE, KB; tools/pmc_traffic.sh); " \
                           "Infinity-Cache hits are counted"
    if rank == 0:
        out = {
            "metric": "image-text pairs/sec fwd+bwd, XFM-base 224px/30tok",
            "value": round(B * world * args.steps / elapsed, 2),
            "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Pretrain.py full multimodal step (ITC+ITM+MLM+MIM), XFM-base, synthetic 224px images + 30-token captions, random init",
                       "pairs_per_gpu": B, "global_batch": B * world, "image_res": 224, "max_tokens": 30,
                       "dropout": not args.eval_mode, "optimizer_step_in_timed_region": not args.no_optimizer,
                       "grad_allreduce": world > 1, "parallelism": f"dp{world}"},
            "step_tflops_per_gpu": round(B * PAIR_GFLOP / ms_per_step, 2),
            "mfma_frac_whole_step": round(B * PAIR_GFLOP / ms_per_step / BF16_DENSE_PEAK_TFLOPS, 4),
            "losses_last_step": loss_vals,
            # the dominant kernel alone (HIP events around its launches in the instrumented step) ...
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_256_kernel<EPI_BF16> (every launch of it in one step: forward / dgrad GEMMs, whole-round parts of tail-split calls included)",
                         "achieved": round(dom_tf, 2), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(dom_tf / BF16_DENSE_PEAK_TFLOPS, 4), "launches": dom_n,
                         "avg_launch_us": round(dom_ms / max(dom_n, 1) * 1e3, 1), "flop_per_launch_avg": dom_fl / max(dom_n, 1),
                         "traffic": traffic, "traffic_note": traffic_note,
                         # ... and the whole gemm_nt family (every forward + dgrad GEMM of the step, small tiles included)
                         "family_gemm_nt": {"achieved": round(achieved, 2), "frac": round(achieved / BF16_DENSE_PEAK_TFLOPS, 4),
                                            "calls": nlaunch, "kernel_ms_per_step": round(gemm_ms, 3), "flop_per_step": gemm_flop}},
        }
        if not args.no_fusion_probe and world == 1:  # single-rank only: its backward would launch unmatched gradient collectives
            out["fusion_encoder_fwd_bwd"] = fusion_probe(model, B, host[0], packed=not args.padded_rows)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only

            out["cpu_baseline"] = cpu_baseline(model, args.cpu_batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
