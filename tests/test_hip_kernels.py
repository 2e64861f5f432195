"""Kernel-level parity on a real MI355X: every C-ABI entry point against plain fp32 PyTorch math on the same inputs.
Tolerances: inputs are bf16 (exact products), accumulation fp32; outputs that are stored as bf16 carry one rounding
(2^-9 relative).  GEMM / attention results are compared at <= 1e-2 of the tensor's max |value| (abs) -- stated per test."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def _fx():
    from xfm_amd import functional as Fx
    return Fx


def _rand(shape, scale=1.0, dtype=BF16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


def _close(got, ref, tol, what=""):
    got, ref = got.float(), ref.float()
    denom = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max()) / denom
    assert err <= tol, f"{what}: max err / max|ref| = {err:.3e} > {tol}"


def gelu_grad(u):
    return 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("M,N,K", [(300, 200, 128), (1920, 768, 768), (788, 2304, 768), (64, 50265, 64), (5, 3, 64)])
@pytest.mark.parametrize("hint", [0, 1, 2, 3, 4, 5, 7, 8])
def test_gemm_nt_bias_bf16_and_f32(M, N, K, hint):
    Fx = _fx()
    a, b = _rand((M, K), seed=1), _rand((N, K), 0.05, seed=2)
    bias = _rand((N,), 0.5, F32, seed=3)
    ref = a.float() @ b.float().t() + bias
    out = Fx.gemm_nt(a, b, bias, tile_hint=hint)
    _close(out, ref, 1e-2, "bf16 out")
    ld = (N + 63) // 64 * 64
    buf = torch.full((M, ld), 7.0, dtype=F32, device="cuda")
    Fx.gemm_nt(a, b, bias, epi=Fx.EPI_F32, out=buf, n=N, tile_hint=hint)
    _close(buf[:, :N], ref, 2e-5, "fp32 out")
    assert float((buf[:, N:] - 7.0).abs().max()) == 0.0 if ld > N else True  # padding columns untouched
    Fx.gemm_nt(a, b, None, epi=Fx.EPI_F32_ACC, out=buf, n=N, tile_hint=hint)
    _close(buf[:, :N], 2 * ref - bias, 2e-5, "fp32 accumulate")


@pytest.mark.parametrize("M,N,K", [(25216, 768, 3072), (25216, 3072, 768), (7680, 2304, 768), (1000, 1000, 192), (70000, 768, 64)])
def test_gemm_nt_tile_configs_agree_bitwise(M, N, K):
    """Every tile config accumulates K in the same order, so outputs must be identical: a race in the pipelined
    256x256 / ring kernels (a fragment read before its direct-to-LDS load landed) shows up as a mismatch."""
    Fx = _fx()
    a, b = _rand((M, K), seed=11), _rand((N, K), 0.05, seed=12)
    ref = Fx.gemm_nt(a, b, tile_hint=1)
    for rep in range(3):
        for hint in (0, 4, 5, 7, 8):  # 0 = auto, incl. the tail split of the N = 768 shapes
            out = Fx.gemm_nt(a, b, tile_hint=hint)
            assert torch.equal(out, ref), f"tile config {hint} rep {rep}: {int((out != ref).sum())} elements differ"


def test_gemm_nt_identity_catches_transposed_maps():
    """A = I with an asymmetric B: any swap in the fragment / accumulator maps shows up exactly."""
    Fx = _fx()
    K = 128
    a = torch.eye(K, dtype=BF16, device="cuda")
    b = (torch.arange(200 * K, device="cuda").reshape(200, K) % 251).to(BF16)  # exact small integers
    for hint in (1, 2, 3, 4, 5, 7, 8):
        out = Fx.gemm_nt(a, b, tile_hint=hint)
        assert torch.equal(out.float(), b.float().t().contiguous()), f"tile config {hint}"


@pytest.mark.parametrize("hint", [0, 4, 5])
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1920, 3072, 768)])
def test_gemm_nt_gelu_and_dgelu(M, N, K, hint):
    Fx = _fx()
    a, b = _rand((M, K), seed=4), _rand((N, K), 0.05, seed=5)
    bias = _rand((N,), 0.5, F32, seed=6)
    pre = a.float() @ b.float().t() + bias
    h, u = Fx.gemm_nt(a, b, bias, epi=Fx.EPI_GELU, tile_hint=hint)
    x = pre.to(BF16).float()  # the GELU sees the bf16-rounded pre-activation (the reference's autocast linear output)
    _close(h, torch.nn.functional.gelu(x), 1e-2, "gelu(pre)")
    _close(u, gelu_grad(x), 1e-2, "gelu'(pre), stored for the backward")
    # dgrad with the GELU derivative folded in: C = (dY . Wt^T) * aux
    dy = _rand((M, N), seed=7)
    wt = _rand((K, N), 0.05, seed=8)  # plays W^T: [K_out, N_contract]
    aux = gelu_grad(_rand((M, K), 1.0, seed=9).float()).to(BF16)
    got = Fx.gemm_nt(dy, wt, epi=Fx.EPI_DGELU, aux=aux, tile_hint=hint)
    ref = (dy.float() @ wt.float().t()) * aux.float()
    _close(got, ref, 1e-2, "dgelu")


@pytest.mark.parametrize("M,N,K", [(25000, 3000, 768), (17000, 1000, 128), (16400, 1001, 64), (66000, 520, 320)])
def test_gemm_nt_persistent_tiles_every_epilogue(M, N, K):
    """More 256 x 256 tiles than CUs: one workgroup per CU walks several tiles, the next tile's staging loads are in flight while
    this tile's epilogue stores go out (counted in the same vmcnt), and the bias comes through LDS.  Ragged M and N (edge tiles take
    the scalar store path and the uncounted wait), K of one to twelve K-tiles, an output stride that is not a multiple of 8 (N = 1001),
    every epilogue, against fp32 math; the one-workgroup-per-tile 128 x 128 kernel must agree bit for bit where the K order is the
    same (bf16 / GELU outputs)."""
    Fx = _fx()
    a, b = _rand((M, K), seed=31), _rand((N, K), 0.05, seed=32)
    bias = _rand((N,), 0.5, F32, seed=33)
    pre = a.float() @ b.float().t() + bias
    out = Fx.gemm_nt(a, b, bias, tile_hint=5)
    _close(out, pre, 1e-2, "bf16 out")
    assert torch.equal(out, Fx.gemm_nt(a, b, bias, tile_hint=1))
    assert torch.equal(Fx.gemm_nt(a, b, None, tile_hint=5), Fx.gemm_nt(a, b, None, tile_hint=1))   # no bias: zeros through LDS
    h, u = Fx.gemm_nt(a, b, bias, epi=Fx.EPI_GELU, tile_hint=5)
    x = pre.to(BF16).float()
    _close(h, torch.nn.functional.gelu(x), 1e-2, "gelu(pre)")
    _close(u, gelu_grad(x), 1e-2, "gelu'(pre)")
    h1, u1 = Fx.gemm_nt(a, b, bias, epi=Fx.EPI_GELU, tile_hint=1)
    assert torch.equal(h, h1) and torch.equal(u, u1)
    aux = gelu_grad(_rand((M, N), 1.0, seed=34).float()).to(BF16)
    got = Fx.gemm_nt(a, b, epi=Fx.EPI_DGELU, aux=aux, tile_hint=5)
    _close(got, (a.float() @ b.float().t()) * aux.float(), 1e-2, "dgelu")
    ld = (N + 63) // 64 * 64
    buf = torch.full((M, ld), 7.0, dtype=F32, device="cuda")
    Fx.gemm_nt(a, b, bias, epi=Fx.EPI_F32, out=buf, n=N, tile_hint=5)
    _close(buf[:, :N], pre, 2e-5, "fp32 out")
    assert ld == N or float((buf[:, N:] - 7.0).abs().max()) == 0.0
    Fx.gemm_nt(a, b, None, epi=Fx.EPI_F32_ACC, out=buf, n=N, tile_hint=5)
    _close(buf[:, :N], 2 * pre - bias, 2e-5, "fp32 accumulate")


@pytest.mark.parametrize("M,N,K,splits", [(788, 200, 136, 0), (1920, 768, 768, 0), (1920, 768, 3072, 3), (64, 136, 64, 1),
                                           (960, 1000, 768, 0),
                                           (4100, 768, 768, 2), (12608, 2304, 768, 5), (5003, 520, 136, 0), (12608, 768, 768, 0), (960, 1000, 136, 1),
                                           (1920, 768, 768, -3), (64, 256, 256, -3), (12608, 2304, 768, -3), (25216, 768, 3072, 0),
                                           (7680, 3072, 768, 0), (12608, 768, 768, -4),
                                           (8250, 768, 3072, 0), (21624, 2304, 768, 0)])   # ragged M: 256 x 256 kernel on M - M % 64 rows + a tail call
def test_gemm_tn_wgrad_accumulates(M, N, K, splits):
    Fx = _fx()
    dy, x = _rand((M, N), seed=10), _rand((M, K), seed=11)
    dw = torch.ones((N, K), dtype=F32, device="cuda")
    db = torch.ones((N,), dtype=F32, device="cuda")
    Fx.gemm_tn(dy, x, dw, splits=splits, dbias=db)
    ref = dy.float().t() @ x.float() + 1.0
    _close(dw, ref, 2e-4, "wgrad")
    _close(db, dy.float().sum(0) + 1.0, 2e-4, "bias grad fused into wgrad")


def test_gemm_tn_exact_integers_catch_layout_errors():
    Fx = _fx()
    M, N, K = 128, 128, 128
    dy = ((torch.arange(M * N, device="cuda").reshape(M, N) * 7) % 13).to(BF16)
    x = ((torch.arange(M * K, device="cuda").reshape(M, K) * 5) % 11).to(BF16)
    dw = torch.zeros((N, K), dtype=F32, device="cuda")
    Fx.gemm_tn(dy, x, dw, splits=1)
    assert torch.equal(dw, dy.float().t() @ x.float())
    # the 256 x 256 pipelined kernel (forced): asymmetric exact-integer operands, two K-tiles
    M, N, K = 128, 256, 512
    dy = ((torch.arange(M * N, device="cuda").reshape(M, N) * 7) % 13).to(BF16)
    x = ((torch.arange(M * K, device="cuda").reshape(M, K) * 5) % 11).to(BF16)
    dw = torch.zeros((N, K), dtype=F32, device="cuda")
    db = torch.zeros((N,), dtype=F32, device="cuda")
    Fx.gemm_tn(dy, x, dw, splits=-3, dbias=db)
    assert torch.equal(dw, dy.float().t() @ x.float())
    assert torch.equal(db, dy.float().sum(0))


@pytest.mark.parametrize("M,N,K,splits", [(5211, 768, 768, 0), (1296, 2304, 768, 0), (83, 128, 256, 1), (2000, 256, 128, 3), (64, 128, 128, 1)])
def test_gemm_tn_ring_kernel_ragged_rows_vs_register_staged_kernel(M, N, K, splits):
    """The 4-slot LDS-ring wgrad kernel (edge-free N, K; any M: packed token rows are not multiples of 32) against fp32 math and
    against the register-staged kernel it replaces (splits = -5 pins the old one), dW += and bias gradient included; exact on
    small-integer operands (layout errors cannot hide)."""
    Fx = _fx()
    dy, x = _rand((M, N), seed=20), _rand((M, K), seed=21)
    dw = torch.full((N, K), 0.5, dtype=F32, device="cuda")
    db = torch.full((N,), 0.25, dtype=F32, device="cuda")
    Fx.gemm_tn(dy, x, dw, splits=splits, dbias=db)
    ref = dy.float().t() @ x.float() + 0.5
    _close(dw, ref, 2e-4, "ring wgrad")
    _close(db, dy.float().sum(0) + 0.25, 2e-4, "ring bias grad")
    dw2 = torch.full((N, K), 0.5, dtype=F32, device="cuda")
    Fx.gemm_tn(dy, x, dw2, splits=-5)
    _close(dw, dw2, 1e-5, "ring vs register-staged")
    dyi = ((torch.arange(M * N, device="cuda").reshape(M, N) * 7) % 13 - 6).to(BF16)
    xi = ((torch.arange(M * K, device="cuda").reshape(M, K) * 5) % 11 - 5).to(BF16)
    dwi = torch.zeros((N, K), dtype=F32, device="cuda")
    dbi = torch.zeros((N,), dtype=F32, device="cuda")
    Fx.gemm_tn(dyi, xi, dwi, splits=1, dbias=dbi)
    assert torch.equal(dwi, dyi.float().t() @ xi.float()) and torch.equal(dbi, dyi.float().sum(0))


def test_cast_transpose_and_colsum():
    Fx = _fx()
    w = _rand((300, 136), dtype=F32, seed=12)
    wb = torch.empty((300, 136), dtype=BF16, device="cuda")
    wt = torch.full((136, 320), 3.0, dtype=BF16, device="cuda")
    Fx.cast_transpose(w, wb, wt)
    assert torch.equal(wb, w.to(BF16))
    assert torch.equal(wt[:, :300], w.to(BF16).t())
    assert float(wt[:, 300:].abs().max()) == 0.0
    y = _rand((1000, 2304), seed=13)
    out = torch.ones(2304, dtype=F32, device="cuda")
    Fx.colsum(y, out)
    _close(out, y.float().sum(0) + 1.0, 1e-5, "colsum")


def test_cast_transpose_batch_matches_single():
    """One launch over a table of weights (ragged shapes, padded leading dimensions, copies that are absent) writes exactly
    what the per-weight launches write."""
    Fx = _fx()
    shapes = [(300, 136, 136, 320), (768, 768, 768, 768), (50, 64, 64, 56), (2304, 768, 768, 2304), (7, 33, 40, 8)]
    entries, want = [], []
    for i, (N, K, ldb, ldt) in enumerate(shapes):
        w = _rand((N, K), dtype=F32, seed=40 + i)
        wb = None if i == 2 else torch.full((N, ldb), 5.0, dtype=BF16, device="cuda")
        wt = None if i == 3 else torch.full((K, ldt), 5.0, dtype=BF16, device="cuda")
        rb = None if wb is None else torch.full((N, ldb), 7.0, dtype=BF16, device="cuda")
        rt = None if wt is None else torch.full((K, ldt), 7.0, dtype=BF16, device="cuda")
        Fx.cast_transpose(w, rb, rt)
        entries.append((w, wb, wt))
        want.append((rb, rt))
    table, n, tiles = Fx.cast_table(entries, torch.device("cuda"))
    Fx.cast_transpose_batch(table, n, tiles)
    for (w, wb, wt), (rb, rt) in zip(entries, want):
        if wb is not None:
            assert torch.equal(wb, rb)
            assert torch.equal(wb[:, :w.shape[1]], w.to(BF16))
        if wt is not None:
            assert torch.equal(wt, rt)


@pytest.mark.parametrize("reps,masked", [(1, True), (2, True), (1, False)])
def test_vit_token_assembly_fwd_bwd(reps, masked):
    """beit2.py:432-446: x * (1 - w) + mask_token * w, then cat(cls, x) -- here for `reps` masked views of the same Bt images."""
    Fx = _fx()
    Bt, P, D = 3, 20, 768
    Bx = Bt * reps
    tok = _rand((Bt, P, D), 1.0, F32, 50).requires_grad_(True)
    cls = _rand((1, 1, D), 1.0, F32, 51).requires_grad_(True)
    mtok = _rand((1, 1, D), 1.0, F32, 52).requires_grad_(True)
    mask = (_rand((Bx, P), 1.0, F32, 53) > 0.3) if masked else None
    if masked:
        mask[0] = False  # a clean view
    x = tok.repeat(reps, 1, 1)
    if masked:
        w = mask.unsqueeze(-1).float()
        x = x * (1 - w) + mtok.expand(Bx, P, -1) * w
    ref = torch.cat([cls.expand(Bx, -1, -1), x], dim=1)
    dx0 = _rand((Bx, P + 1, D), 1.0, F32, 54)
    ref.backward(dx0)
    got = Fx.vit_tokens_fwd(tok.detach(), cls.detach().view(-1), mtok.detach().view(-1), None if mask is None else mask.view(torch.uint8), Bx)
    assert torch.equal(got, ref.detach())
    dcls, dm = torch.ones(D, device="cuda"), torch.ones(D, device="cuda")
    dtok = Fx.vit_tokens_bwd(dx0, None if mask is None else mask.view(torch.uint8), Bt, dcls, dm if masked else None)
    _close(dtok, tok.grad, 1e-6, "dtok")
    _close(dcls, cls.grad.view(-1) + 1.0, 1e-5, "dcls (accumulated)")
    if masked:
        _close(dm, mtok.grad.view(-1) + 1.0, 1e-5, "dmask_token (accumulated)")


@pytest.mark.parametrize("cls_term", [True, False])
def test_mim_loss_fwd_bwd(cls_term):
    """xfm.py:624-635: masked-patch MSE (+ pooled-cls MSE) between the masked and the clean view."""
    Fx = _fx()
    B, N, D = 5, 37, 768
    x = _rand((B, N, D), 1.0, seed=60)
    t = _rand((B, N, D), 1.0, seed=61)
    mask = _rand((B, N - 1), 1.0, F32, 62) > 0.2
    xr = x.float().requires_grad_(True)
    w = mask.unsqueeze(-1).float()
    ref = (((xr[:, 1:] - t.float()[:, 1:]) ** 2) * w).sum() / (w.sum() * D).clamp(min=1.0)
    if cls_term:
        ref = ref + torch.nn.functional.mse_loss(xr[:, 0], t.float()[:, 0])
    (ref * 3.0).backward()
    m8 = mask.contiguous().view(torch.uint8)
    sums = Fx.mim_loss_fwd(x, t, m8)
    got = sums[0] / (sums[2] * D).clamp(min=1.0) + (sums[1] / (B * D) if cls_term else 0.0)
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)), (float(got), float(ref))
    assert float(sums[2]) == float(mask.sum())
    dx = Fx.mim_loss_bwd(x, t, m8, sums, torch.full((1,), 3.0, device="cuda"), cls_term)
    _close(dx, xr.grad, 1e-2, "mim dx")
    assert float(dx[:, 1:][~mask].float().abs().max()) == 0.0  # unmasked patches carry no gradient


def test_pooled_cls_tail_fwd_bwd():
    """beit2.py:455-466: cat([mean over the patch rows, patch rows]) -- in place on the trunk output, and its gradient."""
    Fx = _fx()
    B, N, D = 3, 50, 768
    y = _rand((B * N, D), 1.0, seed=70)
    yr = y.float().view(B, N, D).requires_grad_(True)
    ref = torch.cat([yr[:, 1:].mean(dim=1, keepdim=True), yr[:, 1:]], dim=1)
    dy = _rand((B * N, D), 1.0, seed=71)
    ref.backward(dy.float().view(B, N, D))
    got = Fx.pool_rows_fwd_(y.clone(), B, N).view(B, N, D)
    _close(got[:, 0], ref[:, 0], 1e-2, "pooled row")
    assert torch.equal(got[:, 1:], y.view(B, N, D)[:, 1:])
    g = Fx.pool_rows_bwd(dy, B, N).view(B, N, D)
    _close(g, yr.grad, 1e-2, "pool tail gradient")
    assert float(g[:, 0].float().abs().max()) == 0.0


def _ln_ref(x, w, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


@pytest.mark.parametrize("D", [768, 1536, 3072, 6144])  # 3072 / 6144: the wide-row kernels of the classification head
@pytest.mark.parametrize("dtype", [F32, BF16])
def test_layernorm_plain_fwd_bwd(D, dtype):
    Fx = _fx()
    rows = 333
    x = _rand((rows, D), 2.0, dtype, seed=20)
    w, b = _rand((D,), 1.0, F32, seed=21), _rand((D,), 0.3, F32, seed=22)
    dy = _rand((rows, D), seed=23)
    xr = x.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = _ln_ref(xr, wr, br, 1e-6)
    yr.backward(dy.float())
    y, mean, rstd = Fx.ln_fwd(x, w, b, 1e-6)
    _close(y, yr, 1e-2, "ln fwd")
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dx = torch.empty((rows, D), dtype=F32, device="cuda")
    Fx.ln_bwd(dy, x, mean, rstd, w, dg, db, dx32=dx)
    _close(dx, xr.grad, 1e-4, "ln dx")
    _close(dg, wr.grad, 1e-4, "ln dgamma")
    _close(db, br.grad, 1e-4, "ln dbeta")
    dx2 = torch.ones((rows, D), dtype=F32, device="cuda")
    Fx.ln_bwd(dy, x, mean, rstd, w, dg, db, dx32=dx2, dx_accum=True, dy2=dy)
    _close(dx2, 2 * xr.grad + 1.0, 1e-4, "ln dx accumulate with two dy inputs")


def test_layernorm_post_fwd_bwd_no_dropout():
    Fx = _fx()
    rows, D = 257, 768
    h, res = _rand((rows, D), seed=30), _rand((rows, D), seed=31)
    w, b = _rand((D,), 1.0, F32, seed=32), _rand((D,), 0.3, F32, seed=33)
    dy1, dy2 = _rand((rows, D), seed=34), _rand((rows, D), seed=35)
    hr, rr = h.float().requires_grad_(True), res.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = _ln_ref(hr + rr, wr, br, 1e-5)
    yr.backward(dy1.float() + dy2.float())
    y, z, mean, rstd = Fx.ln_post_fwd(h, res, w, b, 1e-5)
    _close(y, yr, 1e-2, "post-ln fwd")
    _close(z, h.float() + res.float(), 1e-2, "pre-norm sum")
    dg, db, dbias = (torch.zeros(D, device="cuda") for _ in range(3))
    dh, dres = Fx.ln_post_bwd(dy1, z, mean, rstd, w, dg, db, dbias, dy2=dy2)
    assert dh.data_ptr() == dres.data_ptr()
    _close(dh, hr.grad, 2e-2, "post-ln dh")  # z is kept in bf16 for the backward
    _close(dg, wr.grad, 2e-2, "post-ln dgamma")
    _close(db, br.grad, 1e-4, "post-ln dbeta")
    _close(dbias, dh.float().sum(0), 1e-4, "bias grad = colsum(dh)")


def test_layernorm_post_fp32_stream_fwd_bwd():
    """The fp32 residual stream of the text / fusion towers (xroberta._F32_STREAM): the residual is the previous LayerNorm's fp32
    output, z and the twin of y stay fp32, the residual-branch gradient comes in and leaves in fp32 -- fp32-level agreement with torch
    on everything that is not a bf16 tensor by contract (y, dh)."""
    Fx = _fx()
    rows, D = 261, 768
    h, res = _rand((rows, D), seed=30), _rand((rows, D), 1.0, F32, seed=31)
    w, b = _rand((D,), 1.0, F32, seed=32), _rand((D,), 0.3, F32, seed=33)
    dy1, dy2 = _rand((rows, D), seed=34), _rand((rows, D), 1.0, F32, seed=35)
    hr, rr = h.float().requires_grad_(True), res.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = _ln_ref(hr + rr, wr, br, 1e-5)
    yr.backward(dy1.float() + dy2)
    y, z, mean, rstd, y32 = Fx.ln_post_fwd(h, res, w, b, 1e-5, f32=True)
    assert z.dtype == F32 and y32.dtype == F32 and y.dtype == BF16
    _close(z, h.float() + res, 1e-6, "fp32 pre-norm sum")
    _close(y32, yr, 2e-5, "fp32 twin of y")
    assert torch.equal(y, y32.to(BF16)), "y is the rounding of its twin"
    dg, db, dbias = (torch.zeros(D, device="cuda") for _ in range(3))
    dh, dres = Fx.ln_post_bwd(dy1, z, mean, rstd, w, dg, db, dbias, dy2=dy2)
    assert dres.dtype == F32 and dh.dtype == BF16
    _close(dres, rr.grad, 2e-5, "fp32 residual-branch gradient")
    assert torch.equal(dh, dres.to(BF16))
    _close(dg, wr.grad, 1e-4, "dgamma")
    _close(db, br.grad, 1e-4, "dbeta")
    _close(dbias, dh.float().sum(0), 1e-4, "bias grad = colsum(dh)")
    # a bf16 tower input enters the stream through its values; no incoming fp32 gradient on the tower's last LayerNorm
    res16 = res.to(BF16)
    y_b, z_b, m_b, r_b, y32_b = Fx.ln_post_fwd(h, res16, w, b, 1e-5, f32=True)
    _close(z_b, h.float() + res16.float(), 1e-6, "bf16 input, fp32 sum")
    dh_b, dres_b = Fx.ln_post_bwd(dy1, z_b, m_b, r_b, w, torch.zeros_like(dg), torch.zeros_like(db), torch.zeros_like(dbias))
    hr2 = h.float().requires_grad_(True)
    _ln_ref(hr2 + res16.float(), w, b, 1e-5).backward(dy1.float())
    _close(dres_b, hr2.grad, 2e-5, "no fp32 gradient in")
    # dropout: the twin carries the same mask
    drop = Fx.drop_params(0.1, 4242)
    y_d, z_d, _, _, _ = Fx.ln_post_fwd(h, res, w, b, 1e-5, drop, f32=True)
    _, z_d16, _, _ = Fx.ln_post_fwd(h, res16, w, b, 1e-5, drop)
    kept = ((z_d - res).abs() > 0) | (h.float() == 0)
    rate = float(kept.float().mean())
    assert abs(rate - 0.9) < 5e-3, rate
    _close(z_d, z_d16.float() + (res - res16.float()), 2e-2, "same dropout stream as the bf16 form")


def test_layernorm_post_dropout_mask_is_consistent_between_fwd_and_bwd():
    Fx = _fx()
    rows, D, p = 512, 768, 0.1
    h = torch.ones((rows, D), dtype=BF16, device="cuda")
    res = torch.zeros((rows, D), dtype=BF16, device="cuda")
    w, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    drop = Fx.drop_params(p, 12345)
    y, z, mean, rstd = Fx.ln_post_fwd(h, res, w, b, 1e-5, drop)
    keep = z.float() != 0
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p)) < 5e-3, rate
    _close(z.float()[keep], torch.full_like(z.float()[keep], 1 / (1 - p)), 1e-2, "kept values are scaled by 1/(1-p)")
    y2, z2, _, _ = Fx.ln_post_fwd(h, res, w, b, 1e-5, drop)
    assert torch.equal(z, z2), "same seed -> same mask"
    z3 = Fx.ln_post_fwd(h, res, w, b, 1e-5, Fx.drop_params(p, 999))[1]
    assert not torch.equal(z, z3)
    dy = torch.ones((rows, D), dtype=BF16, device="cuda") * torch.arange(D, device="cuda").to(BF16)
    dg, db, dbias = (torch.zeros(D, device="cuda") for _ in range(3))
    dh, dres = Fx.ln_post_bwd(dy, z, mean, rstd, w, dg, db, dbias, drop=drop)
    assert dh.data_ptr() != dres.data_ptr()
    assert torch.equal(dh.float() != 0, keep & (dres.float() != 0)), "backward re-creates the forward mask"
    _close(dh.float()[keep], dres.float()[keep] / (1 - p), 1e-2, "dropout scaling in bwd")


def test_layernorm_layerscale_fwd_bwd():
    Fx = _fx()
    B, N, D = 3, 50, 768
    rows = B * N
    x = _rand((rows, D), 2.0, F32, seed=40)
    h = _rand((rows, D), seed=41)
    g = _rand((D,), 0.1, F32, seed=42)
    rs = torch.tensor([1.0 / 0.9, 0.0, 1.0 / 0.9], device="cuda")
    w, b = _rand((D,), 1.0, F32, seed=43), _rand((D,), 0.3, F32, seed=44)
    dy = _rand((rows, D), seed=45)
    dstream0 = _rand((rows, D), 1.0, F32, seed=46)
    xr, hr, gr = x.clone().requires_grad_(True), h.float().requires_grad_(True), g.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    x1r = xr + rs.repeat_interleave(N).view(-1, 1) * gr * hr
    yr = _ln_ref(x1r, wr, br, 1e-6)
    (yr * dy.float()).sum().backward(retain_graph=True)
    x1r.backward(dstream0)
    x1, y, mean, rstd = Fx.ln_ls_fwd(x, h, g, rs, N, w, b, 1e-6)
    _close(x1, x1r, 1e-6, "stream")
    _close(y, yr, 1e-2, "ls-ln fwd")
    dg, db, dbias, dls = (torch.zeros(D, device="cuda") for _ in range(4))
    dstream = dstream0.clone()
    dh = Fx.ln_ls_bwd(dy, dstream, x1, mean, rstd, w, h, g, rs, N, dg, db, dbias, dls)
    _close(dstream, xr.grad, 1e-4, "stream grad")
    _close(dh, hr.grad, 1e-2, "dh")
    _close(dls, gr.grad, 1e-3, "layer-scale grad")
    _close(dg, wr.grad, 1e-4, "dgamma")
    _close(db, br.grad, 1e-4, "dbeta")
    _close(dbias, dh.float().sum(0), 1e-4, "dbias")


def _attn_ref(q, k, v, B, H, Sq, Sk, scale, bias=None, keep=None, causal=False, pmask=None, pscale=1.0):
    """q [B*Sq, H*64] etc. fp32 leaf tensors -> o [B*Sq, H*64]"""
    qh = q.view(B, Sq, H, 64).permute(0, 2, 1, 3)
    kh = k.view(B, Sk, H, 64).permute(0, 2, 1, 3)
    vh = v.view(B, Sk, H, 64).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias[:, :, :Sk].unsqueeze(0)
    add = torch.zeros(B, 1, Sq, Sk, device=q.device)
    if keep is not None:
        add = add + (1.0 - keep.float())[:, None, None, :] * -10000.0
    if causal:
        cm = torch.ones(Sq, Sk, device=q.device).tril().bool()
        add = torch.where(cm[None, None] & (add == 0), torch.zeros_like(add), torch.full_like(add, -10000.0))
    p = (s + add).softmax(-1)
    if pmask is not None:
        p = p * pmask * pscale
    return (p @ vh).permute(0, 2, 1, 3).reshape(B * Sq, H * 64)


@pytest.mark.parametrize("B,H,Sq,Sk,use_bias,use_keep,causal", [
    (2, 12, 197, 197, True, False, False),   # BEiT self-attention with relative-position bias
    (3, 12, 30, 30, False, True, False),     # text self-attention, ragged padding
    (3, 12, 30, 197, False, True, False),    # cross-attention text -> image tokens
    (2, 4, 30, 30, False, True, True),       # causal decoder
    (1, 2, 130, 300, True, True, False),     # several key chunks, tails on both sides
    (1, 12, 577, 577, True, False, False),   # BEiT at 384 px (BASELINE configs[2]): streamed keys, bias gradient by atomics
    (2, 12, 40, 577, False, True, False),    # cross-attention of 40-token captions onto a 384 px image
    (1, 3, 901, 901, True, False, False),    # 480 px (VQA fine-tuning)
    (1, 1, 2, 3, False, False, False),
    (9, 12, 197, 197, True, False, False),   # short-sequence backward: three batch entries per workgroup (K double buffer, V prefetch)
    (5, 12, 100, 256, True, False, False),   # ... 16 key tiles, 7 query tiles
    (11, 12, 256, 64, False, False, False),  # ... one key tile per key-range wave, no bias
    (3, 12, 50, 20, True, False, False),     # ... two key tiles: two of the four key-range waves idle
    (1, 1, 16, 16, True, False, False),      # ... one tile each way, one batch entry
    (2, 3, 224, 224, True, False, False),    # ... the longest query side the short dK/dV kernel takes (14 query tiles)
    (2, 2, 300, 256, True, False, False),    # ... short dQ kernel (16 key tiles) + general dK/dV kernel (Sq > 224)
    (70, 1, 33, 197, False, False, False),   # ... more batch entries than one round of workgroups: long per-workgroup loops
    (5, 2, 577, 577, True, False, False),    # long-sequence kernels (attention_long.hip): the bias-gradient kernel walks a batch slice
    (3, 2, 300, 700, True, False, False),    # ... Sq != Sk: 2 query blocks x 3 key blocks of 256, ragged tiles on both sides
    (2, 3, 700, 300, True, False, False),    # ... and the other way round
    (2, 2, 577, 577, False, False, False),   # ... without a bias
    (9, 1, 260, 258, True, False, False),    # ... just past the short kernels' reach; nine entries in several slices
])
def test_attention_fwd_bwd(B, H, Sq, Sk, use_bias, use_keep, causal):
    Fx = _fx()
    D = H * 64
    scale = 0.125
    qkv = _rand((B * Sq, D), seed=50)
    kv = _rand((B * Sk, 2 * D), seed=51)
    q, k, v = qkv, kv[:, :D], kv[:, D:]
    ld = (Sk + 15) // 16 * 16
    bias = (_rand((H, Sq, ld), 1.0, F32, seed=52) if use_bias else None)
    keep = None
    if use_keep:
        keep = torch.ones(B, Sk, dtype=torch.int32, device="cuda")
        for i in range(B):
            keep[i, Sk - (i * 7) % max(Sk - 1, 1):] = 0 if i else 1
    dout = _rand((B * Sq, D), seed=53)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    br = bias.clone().requires_grad_(True) if use_bias else None
    ref = _attn_ref(qr, kr, vr, B, H, Sq, Sk, scale, br, keep, causal)
    ref.backward(dout.float())
    o, lse = Fx.attn_fwd(q, k, v, B, H, Sq, Sk, scale, bias=bias, key_keep=keep, causal=causal)
    _close(o, ref, 1e-2, "attention out")
    dq = torch.empty((B * Sq, D), dtype=BF16, device="cuda")
    dkv = torch.empty((B * Sk, 2 * D), dtype=BF16, device="cuda")
    dbias = torch.zeros_like(bias) if use_bias else None
    Fx.attn_bwd(dout, q, k, v, o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Sq, Sk, scale, bias=bias, dbias=dbias, key_keep=keep,
                causal=causal)
    _close(dq, qr.grad, 2e-2, "dq")
    _close(dkv[:, :D], kr.grad, 2e-2, "dk")
    _close(dkv[:, D:], vr.grad, 2e-2, "dv")
    if use_bias:
        _close(dbias[:, :, :Sk], br.grad[:, :, :Sk], 2e-2, "dbias")
        # same backward with the transposed bias copy (vector loads in the dK/dV kernel)
        ldt = (Sq + 15) // 16 * 16
        bias_t = torch.zeros((H, Sk, ldt), dtype=F32, device="cuda")
        bias_t[:, :, :Sq] = bias[:, :, :Sk].transpose(1, 2)
        dq2, dkv2 = torch.empty_like(dq), torch.empty_like(dkv)
        Fx.attn_bwd(dout, q, k, v, o, lse, dq2, dkv2[:, :D], dkv2[:, D:], B, H, Sq, Sk, scale, bias=bias, dbias=torch.zeros_like(bias),
                    key_keep=keep, causal=causal, bias_t=bias_t)
        _close(dkv2[:, :D], kr.grad, 2e-2, "dk (bias_t)")
        _close(dkv2[:, D:], vr.grad, 2e-2, "dv (bias_t)")


@pytest.mark.parametrize("B,H,Sq,Sk,use_bias,use_keep,drop_p", [(3, 12, 197, 197, True, False, 0.0), (2, 4, 577, 577, True, False, 0.0),
                                                                  (5, 12, 30, 30, False, True, 0.0), (5, 12, 30, 30, False, True, 0.1),
                                                                  (4, 12, 30, 197, False, False, 0.1)])
def test_attention_backward_single_pass_delta_vs_two_pass(B, H, Sq, Sk, use_bias, use_keep, drop_p):
    """delta_i = sum_j P_ij dP_ij taken from dO . (o + o_lo) (forward output kept as bf16 hi + lo halves) against the exact first
    pass over the keys: gradients agree to well below the bf16 noise of dS, dropout included (the identity holds for the dropped
    probabilities too), and both agree with fp32 math."""
    Fx = _fx()
    D = H * 64
    scale = 0.125
    q, kv = _rand((B * Sq, D), seed=70), _rand((B * Sk, 2 * D), seed=71)
    k, v = kv[:, :D], kv[:, D:]
    ld = (Sk + 15) // 16 * 16
    bias = _rand((H, Sq, ld), 1.0, F32, seed=72) if use_bias else None
    keep = None
    if use_keep:
        keep = torch.ones(B, Sk, dtype=torch.int32, device="cuda")
        for i in range(1, B):
            keep[i, Sk - (i * 7) % (Sk - 1):] = 0
    drop = Fx.drop_params(drop_p, 4242)
    dout = _rand((B * Sq, D), seed=73)
    o, lse, o_lo = Fx.attn_fwd(q, k, v, B, H, Sq, Sk, scale, bias=bias, key_keep=keep, drop=drop, lo=True)
    o2, _ = Fx.attn_fwd(q, k, v, B, H, Sq, Sk, scale, bias=bias, key_keep=keep, drop=drop)
    assert torch.equal(o, o2)
    assert float(o_lo.float().abs().max()) <= float(o.float().abs().max()) * 2 ** -8   # what bf16 lost: below half an ulp of o

    def bwd(lo):
        dq = torch.empty((B * Sq, D), dtype=BF16, device="cuda")
        dkv = torch.empty((B * Sk, 2 * D), dtype=BF16, device="cuda")
        dbias = torch.zeros_like(bias) if use_bias else None
        Fx.attn_bwd(dout, q, k, v, o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Sq, Sk, scale, bias=bias, dbias=dbias, key_keep=keep,
                    drop=drop, o_lo=lo)
        return dq, dkv, dbias

    dq_a, dkv_a, db_a = bwd(None)
    dq_b, dkv_b, db_b = bwd(o_lo)
    for name, x, y in (("dq", dq_b, dq_a), ("dkv", dkv_b, dkv_a)) + ((("dbias", db_b, db_a),) if use_bias else ()):
        rel = float((x.float() - y.float()).norm() / (y.float().norm() + 1e-30))
        assert rel <= 6e-3, (name, rel)
    if drop_p == 0.0:
        qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
        br = bias.clone().requires_grad_(True) if use_bias else None
        ref = _attn_ref(qr, kr, vr, B, H, Sq, Sk, scale, br, keep, False)
        ref.backward(dout.float())
        _close(dq_b, qr.grad, 2e-2, "dq (single pass)")
        _close(dkv_b[:, :D], kr.grad, 2e-2, "dk (single pass)")
        _close(dkv_b[:, D:], vr.grad, 2e-2, "dv (single pass)")


def test_attention_dropout_mask_consistency():
    """V = I exposes the dropped probabilities: O = P_dropped.  The same mask must drive the backward."""
    Fx = _fx()
    B, H, Sq, Sk, p = 2, 2, 48, 64, 0.1
    D = H * 64
    q, k = _rand((B * Sq, D), 0.3, seed=60), _rand((B * Sk, D), 0.3, seed=61)
    v = torch.eye(64, dtype=BF16, device="cuda").repeat(B, H).contiguous()  # [B*64, H*64]: per (b,h) identity
    drop = Fx.drop_params(p, 777)
    o, lse = Fx.attn_fwd(q, k, v, B, H, Sq, Sk, 0.125, drop=drop)
    o0, _ = Fx.attn_fwd(q, k, v, B, H, Sq, Sk, 0.125)
    pm = (o.float() != 0)
    rate = float(pm.float().mean())
    assert abs(rate - (1 - p)) < 2e-2, rate
    _close(o.float()[pm], (o0.float() / (1 - p))[pm], 2e-2, "kept probabilities scaled")
    pmask = pm.view(B, Sq, H, 64).permute(0, 2, 1, 3).float()
    dout = _rand((B * Sq, D), seed=62)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, B, H, Sq, Sk, 0.125, pmask=pmask, pscale=1 / (1 - p))
    ref.backward(dout.float())
    dq, dk, dv = (torch.empty_like(t) for t in (q, k, v))
    Fx.attn_bwd(dout, q, k, v, o, lse, dq, dk, dv, B, H, Sq, Sk, 0.125, drop=drop)
    _close(dq, qr.grad, 3e-2, "dq with dropout")
    _close(dk, kr.grad, 3e-2, "dk with dropout")
    _close(dv, vr.grad, 3e-2, "dv with dropout")


def test_relpos_gather_scatter():
    Fx = _fx()
    from oracle.xfm_oracle import beit_relative_position_index
    H, g = 12, 14
    N = g * g + 1
    idx = beit_relative_position_index(g, g).cuda()
    table = _rand(((2 * g - 1) ** 2 + 3, H), 1.0, F32, seed=70)
    ld = 208
    dense = Fx.relpos_gather(table, idx.to(torch.int32).contiguous(), H, N, ld)
    ref = table[idx.view(-1)].view(N, N, H).permute(2, 0, 1)
    assert torch.equal(dense[:, :, :N], ref)
    assert float(dense[:, :, N:].abs().max()) == 0.0
    dd = _rand((H, N, ld), 1.0, F32, seed=71)
    dt = torch.zeros_like(table)
    Fx.relpos_scatter(dd, idx.to(torch.int32).contiguous(), H, N, ld, dt)
    tr = table.clone().requires_grad_(True)
    (tr[idx.view(-1)].view(N, N, H).permute(2, 0, 1) * dd[:, :, :N]).sum().backward()
    _close(dt, tr.grad, 1e-5, "relpos table grad")
    # atomic-free form: positions pre-sorted by table entry, accumulates (+=) into the gradient
    order, start = Fx.relpos_sorted_index(idx.to(torch.int32).contiguous(), table.shape[0])
    dt2 = torch.ones_like(table)
    Fx.relpos_scatter_sorted(dd, order, start, H, N, ld, dt2)
    _close(dt2 - 1.0, tr.grad, 1e-5, "relpos table grad (sorted gather)")


def test_patchify_matches_conv_unfold():
    Fx = _fx()
    img = _rand((3, 3, 224, 224), 1.0, F32, seed=80)
    got = Fx.patchify(img, 16)
    ref = img.view(3, 3, 14, 16, 14, 16).permute(0, 2, 4, 1, 3, 5).reshape(3 * 196, 768).to(BF16)
    assert torch.equal(got, ref)


def test_embedding_layernorm_fwd_bwd():
    Fx = _fx()
    from oracle.xfm_oracle import roberta_position_ids
    from xfm_amd import synthetic as syn
    V, D, B, T = 1000, 768, 5, 30
    b = syn.pretrain_batch(B, seed=3, with_image=False, vocab=V)
    ids = b["text_ids"].cuda()
    word, pos, typ = _rand((V, D), 0.1, F32, 90), _rand((64, D), 0.1, F32, 91), _rand((1, D), 0.1, F32, 92)
    w, bb = _rand((D,), 1.0, F32, 93), _rand((D,), 0.3, F32, 94)
    dy = _rand((B * T, D), seed=95)
    wr, pr, tr_, lw, lb = (t.clone().requires_grad_(True) for t in (word, pos, typ, w, bb))
    pid = roberta_position_ids(ids.cpu(), 1).cuda()
    e = torch.nn.functional.embedding(ids, wr, padding_idx=1) + tr_[0] + torch.nn.functional.embedding(pid, pr, padding_idx=1)
    ref = _ln_ref(e, lw, lb, 1e-5).view(B * T, D)
    ref.backward(dy.float())
    y, mean, rstd, pos_ids = Fx.embed_ln_fwd(ids, word, pos, typ, w, bb, 1e-5, 1)
    assert torch.equal(pos_ids.view(B, T).long(), pid)
    _close(y, ref, 1e-2, "embedding fwd")
    dword, dpos, dtyp, dg, db = (torch.zeros_like(t) for t in (word, pos, typ.view(-1), w, bb))
    Fx.embed_ln_bwd(dy, ids, word, pos, typ, w, bb, 1e-5, 1, mean, rstd, pos_ids, dword, dpos, dtyp, dg, db)
    _close(dword, wr.grad, 1e-4, "dword")
    _close(dpos, pr.grad, 1e-4, "dpos")
    _close(dtyp, tr_.grad.view(-1), 1e-4, "dtype")
    _close(dg, lw.grad, 1e-4, "dgamma")
    _close(db, lb.grad, 1e-4, "dbeta")
    assert float(dword[1].abs().max()) == 0.0 and float(dpos[1].abs().max()) == 0.0, "padding rows take no gradient"


def test_embedding_fp32_twin_and_fp32_gradient():
    """xfm_embed_args.y32 / dy32: the embedding output's fp32 twin is the un-rounded value of y, and an fp32 gradient on the twin's edge is
    added to the bf16 one."""
    Fx = _fx()
    from xfm_amd import synthetic as syn
    V, D, B, T = 1000, 768, 5, 30
    b = syn.pretrain_batch(B, seed=3, with_image=False, vocab=V)
    ids = b["text_ids"].cuda()
    word, pos, typ = _rand((V, D), 0.1, F32, 90), _rand((64, D), 0.1, F32, 91), _rand((1, D), 0.1, F32, 92)
    w, bb = _rand((D,), 1.0, F32, 93), _rand((D,), 0.3, F32, 94)
    y, mean, rstd, pos_ids, y32 = Fx.embed_ln_fwd(ids, word, pos, typ, w, bb, 1e-5, 1, y32=True)
    y0 = Fx.embed_ln_fwd(ids, word, pos, typ, w, bb, 1e-5, 1)[0]
    assert torch.equal(y, y0) and torch.equal(y, y32.to(BF16))
    dy, dy32 = _rand((B * T, D), seed=95), _rand((B * T, D), 1.0, F32, seed=96)
    both = [torch.zeros_like(t) for t in (word, pos, typ.view(-1), w, bb)]
    Fx.embed_ln_bwd(dy, ids, word, pos, typ, w, bb, 1e-5, 1, mean, rstd, pos_ids, *both, dy32=dy32)
    parts = [torch.zeros_like(t) for t in (word, pos, typ.view(-1), w, bb)]
    Fx.embed_ln_bwd(dy, ids, word, pos, typ, w, bb, 1e-5, 1, mean, rstd, pos_ids, *parts)
    zero16 = torch.zeros_like(dy)
    hi = dy32.to(BF16)
    lo = (dy32 - hi.float()).to(BF16)      # dy32 ~ hi + lo: two more bf16-gradient calls reproduce the fp32 edge to 2^-16
    for extra in (hi, lo):
        Fx.embed_ln_bwd(extra, ids, word, pos, typ, w, bb, 1e-5, 1, mean, rstd, pos_ids, *parts)
    del zero16
    for got, ref, name in zip(both, parts, ("dword", "dpos", "dtype", "dgamma", "dbeta")):
        _close(got, ref, 2e-4, name)


def test_embedding_gradient_ordered_scatter_is_bit_reproducible(monkeypatch):
    """XFM_DETERMINISTIC=1 (xfm_embed_args.dz_out + xfm_rows_segment_sum): the per-token gradient is stored and added to the word / position
    tables in sorted runs, one owner per table row and rows in a fixed order -- the same sums as the float-atomic scatter to rounding, the
    pad row and skipped tokens excluded alike, and bit-identical from call to call (the atomic form is not)."""
    Fx = _fx()
    from xfm_amd import synthetic as syn
    V, D, B, T = 300, 768, 48, 30            # a small vocabulary: long runs of equal ids
    b = syn.pretrain_batch(B, seed=5, with_image=False, vocab=V)
    ids = b["text_ids"].cuda()
    word, pos, typ = _rand((V, D), 0.1, F32, 190), _rand((64, D), 0.1, F32, 191), _rand((1, D), 0.1, F32, 192)
    w, bb = _rand((D,), 1.0, F32, 193), _rand((D,), 0.3, F32, 194)
    y, mean, rstd, pos_ids = Fx.embed_ln_fwd(ids, word, pos, typ, w, bb, 1e-5, 1)
    dy = _rand((B * T, D), seed=195)

    def run():
        out = [torch.zeros_like(t) for t in (word, pos, typ.view(-1), w, bb)]
        Fx.embed_ln_bwd(dy, ids, word, pos, typ, w, bb, 1e-5, 1, mean, rstd, pos_ids, *out)
        torch.cuda.synchronize()
        return out

    monkeypatch.setenv("XFM_DETERMINISTIC", "0")
    atomic = run()
    monkeypatch.setenv("XFM_DETERMINISTIC", "1")
    first = run()
    for _ in range(3):
        again = run()
        for a_, b_ in zip(first, again):
            assert torch.equal(a_, b_)
    for got, ref, name in zip(first, atomic, ("dword", "dpos", "dtype", "dgamma", "dbeta")):
        _close(got, ref, 1e-5, name)
    assert float(first[0][1].abs().max()) == 0.0 and float(first[1][1].abs().max()) == 0.0   # the pad rows (id 1) take no gradient
    # the segment sum on its own against index_add_
    src = _rand((500, 64), 1.0, F32, 196)
    keys = torch.randint(-1, 20, (500,), generator=torch.Generator().manual_seed(2)).cuda()
    skey, perm = torch.sort(keys, stable=True)
    out = torch.zeros((20, 64), dtype=F32, device="cuda")
    from xfm_amd import _lib
    from xfm_amd._lib import check
    check(_lib.load().xfm_rows_segment_sum(src.data_ptr(), perm.data_ptr(), skey.data_ptr(), 500, 64, 7, out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "rows_segment_sum")
    keep = (keys >= 0) & (keys != 7)
    ref = torch.zeros_like(out).index_add_(0, keys[keep], src[keep])
    _close(out, ref, 1e-6, "segment sum")
    assert float(out[7].abs().max()) == 0.0


def test_cross_entropy_fwd_bwd_ignore_index():
    Fx = _fx()
    R, V, ld = 37, 50265, 50304
    logits = torch.zeros((R, ld), dtype=F32, device="cuda")
    logits[:, :V] = _rand((R, V), 2.0, F32, seed=100)
    logits[:, V:] = 1e9  # padding columns must be ignored
    labels = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(1)).cuda()
    labels[::5] = -100
    lr = logits[:, :V].clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, labels, ignore_index=-100)
    ref.backward()
    lse, loss_rows = Fx.ce_fwd(logits, V, labels)
    nvalid = (labels != -100).sum()
    assert abs(float(loss_rows.sum() / nvalid) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    scale = (1.0 / nvalid.float()).reshape(1)
    d = Fx.ce_bwd(logits, V, labels, lse, scale, ld)
    _close(d[:, :V], lr.grad, 1e-2, "dlogits")
    assert float(d[:, V:].float().abs().max()) == 0.0 and float(d[::5].float().abs().max()) == 0.0


def test_cross_entropy_bwd_per_row_scale():
    """reduction='none' (RobertaForCausalLM, xroberta.py:1107-1110): every row carries its own upstream gradient."""
    Fx = _fx()
    R, V, ld = 23, 1000, 1024
    logits = torch.zeros((R, ld), dtype=F32, device="cuda")
    logits[:, :V] = _rand((R, V), 2.0, F32, seed=101)
    labels = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(2)).cuda()
    labels[::4] = -100
    w = _rand((R,), 1.0, F32, seed=102)
    lr = logits[:, :V].clone().requires_grad_(True)
    ref_rows = torch.nn.functional.cross_entropy(lr, labels, ignore_index=-100, reduction="none")
    (ref_rows * w).sum().backward()
    lse, loss_rows = Fx.ce_fwd(logits, V, labels)
    _close(loss_rows, ref_rows.detach(), 1e-5, "loss rows")
    d = Fx.ce_bwd(logits, V, labels, lse, w.contiguous(), ld)
    _close(d[:, :V], lr.grad, 1e-2, "dlogits")
    assert float(d[::4].float().abs().max()) == 0.0


def test_adamw_and_sumsq_flat_arena():
    Fx = _fx()
    n = 256 * 40
    p0 = _rand((n,), 1.0, F32, 110)
    g = _rand((n,), 0.1, F32, 111)
    group = (torch.arange(n // 256) % 2).to(torch.uint8).cuda()
    lrs, wds = [1e-3, 2e-3], [0.01, 0.0]
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ss = torch.zeros(1, device="cuda")
    Fx.sumsq(g, ss)
    _close(ss, (g.double() ** 2).sum().float().view(1), 1e-5, "sumsq")
    # bit-reproducible (fixed-order partials, no atomics): data-parallel ranks derive the clip coefficient from it independently
    big = _rand((1024 * 1024 * 8 + 4,), 0.1, F32, 112)
    outs = []
    for _ in range(3):
        o = torch.zeros(1, device="cuda")
        Fx.sumsq(big, o)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    _close(outs[0], (big.double() ** 2).sum().float().view(1), 1e-5, "sumsq (grid-capped)")
    Fx.sumsq(g, outs[0])  # accumulates into out
    _close(outs[0], ((big.double() ** 2).sum() + (g.double() ** 2).sum()).float().view(1), 1e-5, "sumsq accumulate")
    clip = torch.tensor([0.5], device="cuda")
    for step in (1, 2):
        Fx.adamw(p, g, m, v, group, lrs, wds, 0.9, 0.98, 1e-8, step, clip)
    # zero_grad in the same sweep: same update, gradient left at exactly zero
    p2, m2, v2, g2 = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), g.clone()
    Fx.adamw(p2, g2, m2, v2, group, lrs, wds, 0.9, 0.98, 1e-8, 1, clip, zero_grad=True)
    p3, m3, v3 = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    Fx.adamw(p3, g, m3, v3, group, lrs, wds, 0.9, 0.98, 1e-8, 1, clip)
    assert torch.equal(p2, p3) and torch.equal(m2, m3) and torch.equal(v2, v3) and float(g2.abs().max()) == 0.0 and float(g.abs().max()) > 0
    ref = p0.clone().requires_grad_(True)
    mask = (group.repeat_interleave(256) == 0)
    ref_a, ref_b = ref[mask].detach().clone().requires_grad_(True), ref[~mask].detach().clone().requires_grad_(True)
    opt = torch.optim.AdamW([{"params": [ref_a], "lr": lrs[0], "weight_decay": wds[0]},
                             {"params": [ref_b], "lr": lrs[1], "weight_decay": wds[1]}], betas=(0.9, 0.98), eps=1e-8)
    for step in (1, 2):
        ref_a.grad, ref_b.grad = g[mask] * 0.5, g[~mask] * 0.5
        opt.step()
    # transformers' AdamW applies the decay after the Adam update, torch's before: identical to first order in lr*wd
    _close(p[mask], ref_a, 1e-4, "adamw group 0")
    _close(p[~mask], ref_b, 1e-4, "adamw group 1")


def test_attention_shared_kv_sources_and_row_fold():
    """kv_index: several query batch rows read one key/value source; folding the per-row dK/dV equals attending to gathered copies."""
    Fx = _fx()
    B, U, H, Sq, Sk = 6, 3, 4, 30, 197
    D = H * 64
    idx = torch.tensor([0, 2, 1, 0, 0, 2], dtype=torch.int32, device="cuda")
    q = _rand((B * Sq, D), seed=200)
    kv = _rand((U * Sk, 2 * D), seed=201)
    keep = torch.ones(U, Sk, dtype=torch.int32, device="cuda")
    keep[1, 150:] = 0
    dout = _rand((B * Sq, D), seed=202)
    # reference: gather K/V per query row
    kvg = kv.view(U, Sk, 2 * D)[idx.long()].reshape(B * Sk, 2 * D).contiguous()
    keepg = keep[idx.long()].contiguous()
    o_ref, lse_ref = Fx.attn_fwd(q, kvg[:, :D], kvg[:, D:], B, H, Sq, Sk, 0.125, key_keep=keepg)
    o, lse = Fx.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, Sq, Sk, 0.125, key_keep=keep, kv_index=idx)
    assert torch.equal(o, o_ref) and torch.equal(lse[..., :Sq], lse_ref[..., :Sq])  # lse rows are padded to a multiple of 4
    dq_ref, dkv_ref = torch.empty_like(q), torch.empty_like(kvg)
    Fx.attn_bwd(dout, q, kvg[:, :D], kvg[:, D:], o_ref, lse_ref, dq_ref, dkv_ref[:, :D], dkv_ref[:, D:], B, H, Sq, Sk, 0.125, key_keep=keepg)
    dq, dkv = torch.empty_like(q), torch.empty((B * Sk, 2 * D), dtype=BF16, device="cuda")
    Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Sq, Sk, 0.125, key_keep=keep, kv_index=idx)
    assert torch.equal(dq, dq_ref)
    assert not bool(torch.isnan(dkv.float()).any())
    _close(dkv, dkv_ref, 1e-6, "per-row dK/dV with shared sources")
    folded = Fx.rows_index_sum(dkv, idx, U, Sk)
    ref = torch.zeros(U, Sk * 2 * D, device="cuda").index_add_(0, idx.long(), dkv_ref.float().view(B, -1)).view(U * Sk, 2 * D)
    _close(folded, ref, 1e-2, "folded dK/dV")


@pytest.mark.parametrize("B,U,Sq,Sk,p", [(6, 3, 30, 197, 0.0), (11, 4, 30, 197, 0.1), (5, 5, 40, 64, 0.1), (9, 2, 64, 256, 0.0),
                                         (11, 4, 40, 577, 0.1),    # 384-px retrieval: streamed keys (10 chunks through a two-slot ring)
                                         (7, 3, 40, 901, 0.0),     # 480-px VQA: 15 chunks, the last one 5 keys
                                         (30, 2, 40, 300, 0.1)])   # ~15 rows x 3 tiles per image: three passes of 16 tile slots
@pytest.mark.parametrize("masked", [True, False])
def test_attention_grouped_by_kv_source(B, U, Sq, Sk, p, masked):
    """Grouped mode (one workgroup per key/value source and head, dK/dV summed over the group's rows in registers; more than 256 keys
    stream through LDS) against the kv_index path + row fold: same masks, same dropout stream (keyed by the query batch row), empty
    groups give zero dK/dV."""
    Fx = _fx()
    H = 4
    D = H * 64
    gen = torch.Generator().manual_seed(5)
    idx = torch.randint(0, max(U - 1, 1), (B,), generator=gen).to(torch.int32).cuda()  # the last source stays unused when U > 1
    q = _rand((B * Sq, D), seed=210)
    kv = _rand((U * Sk, 2 * D), seed=211)
    keep = None   # masked = False: the instantiations without key-keep flags (the packed fusion tower's case)
    if masked:
        keep = torch.ones(U, Sk, dtype=torch.int32, device="cuda")
        keep[0, Sk - 20:] = 0
    dout = _rand((B * Sq, D), seed=212)
    drop = Fx.drop_params(p, 1234567)
    o_ref, lse_ref = Fx.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, Sq, Sk, 0.125, key_keep=keep, kv_index=idx, drop=drop)
    groups = Fx.kv_groups(idx, U)
    o, lse = Fx.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, Sq, Sk, 0.125, key_keep=keep, groups=groups, drop=drop)
    _close(o, o_ref, 4e-3, "grouped forward")   # exponent-of-2 arithmetic: equal to fp32 rounding, one bf16 ulp after the output rounding
    assert float((o.float() - o_ref.float()).abs().mean()) <= 2e-4 * float(o_ref.float().abs().max()), "grouped forward: mean deviation"
    _close(lse[..., :Sq], lse_ref[..., :Sq], 2e-6, "grouped lse")
    dq_ref, dkv_rows = torch.empty_like(q), torch.empty((B * Sk, 2 * D), dtype=BF16, device="cuda")
    Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o_ref, lse_ref, dq_ref, dkv_rows[:, :D], dkv_rows[:, D:], B, H, Sq, Sk, 0.125,
                key_keep=keep, kv_index=idx, drop=drop)
    ref = torch.zeros(U, Sk * 2 * D, device="cuda").index_add_(0, idx.long(), dkv_rows.float().view(B, -1)).view(U * Sk, 2 * D)
    dq, dkv = torch.empty_like(q), torch.full((U * Sk, 2 * D), 7.0, dtype=BF16, device="cuda")
    Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Sq, Sk, 0.125, key_keep=keep, groups=groups, drop=drop)
    # (the grouped kernels take the probabilities in the exponent of 2 -- exp2(s c log2e - lse log2e) against exp(s c - lse) -- and the
    # streamed ones sum the chunks' dQ in a different association: the fp32 values agree to rounding, the bf16 results to one ulp of the
    # largest entry)
    _close(dq, dq_ref, 4e-3, "grouped dQ")
    assert float((dq.float() - dq_ref.float()).abs().mean()) <= 2e-4 * float(dq_ref.float().abs().max()), "grouped dQ: mean deviation"
    _close(dkv, ref, 1e-2, "grouped dK/dV (summed per source)")
    if U > 1:
        assert float(dkv.view(U, -1)[U - 1].float().abs().max()) == 0.0, "unused source must get zero gradients"
    # one-sweep form (what the layer executor runs): the forward also leaves the low half of O, delta = dO . (O + Olo)
    o2, lse2, o_lo = Fx.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, Sq, Sk, 0.125, key_keep=keep, groups=groups, drop=drop, lo=True)
    assert torch.equal(o2, o) and torch.equal(lse2[groups[1].long()][..., :Sq], lse[groups[1].long()][..., :Sq])
    dq1, dkv1 = torch.empty_like(q), torch.full((U * Sk, 2 * D), 7.0, dtype=BF16, device="cuda")
    d1 = Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o, lse, dq1, dkv1[:, :D], dkv1[:, D:], B, H, Sq, Sk, 0.125, key_keep=keep, groups=groups,
                     drop=drop, o_lo=o_lo)
    d0 = Fx.attn_bwd(dout, q, kv[:, :D], kv[:, D:], o, lse, dq, dkv[:, :D], dkv[:, D:], B, H, Sq, Sk, 0.125, key_keep=keep, groups=groups, drop=drop)
    rows = groups[1].long()   # (rows of empty trailing sources never appear; every listed row has its statistics written)
    # (dO . O carries the bf16 rounding of the probabilities that went into the P V product; the two-sweep sum uses them in fp32)
    _close(d1[rows][..., :Sq], d0[rows][..., :Sq], 4e-3, "delta from the output halves vs the two-sweep sum")
    for name, x, y in (("dq", dq1, dq), ("dkv", dkv1, dkv)):   # the bound of test_attention_backward_single_pass_delta_vs_two_pass
        rel = float((x.float() - y.float()).norm() / (y.float().norm() + 1e-30))
        assert rel <= 6e-3, (name, rel)
    _close(dq1, dq_ref, 1e-2, "grouped dQ, one sweep")
    _close(dkv1, ref, 1e-2, "grouped dK/dV, one sweep")


def test_gemm_nt_split_k_accumulate():
    """Few output tiles, very long K, fp32-accumulate epilogue: the launcher slices K over gridDim.y (atomics into C)."""
    Fx = _fx()
    M, N, K = 960, 768, 50304
    a, b = _rand((M, K), 0.05, seed=31), _rand((N, K), 0.05, seed=32)
    bias = _rand((N,), 0.5, F32, seed=33)
    c = torch.full((M, N), 2.0, dtype=F32, device="cuda")
    Fx.gemm_nt(a, b, bias, epi=Fx.EPI_F32_ACC, out=c)
    ref = a.float() @ b.float().t() + bias + 2.0
    _close(c, ref, 1e-4, "split-K fp32 accumulate")


@pytest.mark.parametrize("M,N,K", [(960, 768, 50304), (60, 768, 50304), (225, 768, 30528), (960, 768, 4096)])
def test_gemm_nt_ksplit_is_exact_and_reproducible(M, N, K):
    """The LM-head activation gradient (xroberta.py:1325-1333): K sliced over the grid, partial planes summed in slice order.  Values
    vs fp32 math (bf16 and fp32 outputs, with and without a bias), and -- the point of it -- the same bits on every launch: the
    atomics merge it replaced changed the last bits of the sum from run to run, which the bf16 rounding that follows turns into
    whole-ulp differences of an activation gradient (tools/cold_probe.py).  (960, 768, 4096) takes the unsliced plan."""
    Fx = _fx()
    a, b = _rand((M, K), 0.05, seed=41), _rand((N, K), 0.05, seed=42)
    bias = _rand((N,), 0.5, F32, seed=43)
    ref = a.float() @ b.float().t()
    o32 = Fx.gemm_nt_ksplit(a, b, out_dtype=F32)
    _close(o32, ref, 1e-4, "k-sliced fp32")
    _close(Fx.gemm_nt_ksplit(a, b, bias=bias, out_dtype=F32), ref + bias, 1e-4, "k-sliced fp32 + bias")
    o16 = Fx.gemm_nt_ksplit(a, b)
    assert o16.dtype == BF16
    if Fx._lib.load().xfm_gemm_nt_ksplit_workspace(M, N, K) > 0:   # sliced: ONE rounding of the fp32 sum
        assert torch.equal(o16, o32.to(BF16))
    else:
        _close(o16, ref, 1e-2, "unsliced bf16")
    for _ in range(5):
        junk = torch.empty(64 << 20, dtype=torch.uint8, device="cuda").random_(0, 255)   # other traffic between the launches
        del junk
        assert torch.equal(Fx.gemm_nt_ksplit(a, b), o16) and torch.equal(Fx.gemm_nt_ksplit(a, b, out_dtype=F32), o32)


# ---- contrastive / matching glue (xfm.py:614-621, 683-746) ---------------------------------------------------------------------
@pytest.mark.parametrize("R,E", [(64, 256), (7, 768), (300, 64)])
def test_rownorm_matches_normalize(R, E):
    from xfm_amd.ops import row_normalize
    x = _rand((R, E), 3.0, F32, seed=1).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    dy = _rand((R, E), 1.0, F32, seed=2)
    y = row_normalize(x)
    ref = torch.nn.functional.normalize(xr, dim=-1)
    y.backward(dy)
    ref.backward(dy)
    _close(y, ref, 1e-6, "normalize")
    _close(x.grad, xr.grad, 1e-5, "normalize backward")


# 3072 = 128 x 24 GPUs, the global batch the reference pre-trains at (the gathered rows live in dynamic LDS: no 2048-row cap)
@pytest.mark.parametrize("N,E", [(64, 256), (5, 256), (512, 256), (130, 128), (3072, 256)])
@pytest.mark.parametrize("temp", [0.07, 0.5])
def test_itc_loss_matches_two_cross_entropies(N, E, temp):
    """fp32 in, fp32 out: 1e-5 relative on the loss, 1e-4 of max |grad| on the gradients (fast-math exp / log in the kernel)."""
    from xfm_amd.ops import itc_loss
    F = torch.nn.functional
    I = F.normalize(_rand((N, E), 1.0, F32, seed=1), dim=-1).requires_grad_(True)
    T = F.normalize(_rand((N, E), 1.0, F32, seed=2) + 0.5 * I.detach(), dim=-1).requires_grad_(True)
    t = torch.tensor(temp, device="cuda", requires_grad=True)
    Ir, Tr, tr = (v.detach().double().requires_grad_(True) for v in (I, T, t))
    loss = itc_loss(I, T, t)
    logits = Ir @ Tr.t() / tr
    labels = torch.arange(N, device="cuda")
    ref = (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2
    (loss * 1.7).backward()
    (ref * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(abs(float(ref)), 1.0)
    _close(I.grad, Ir.grad, 1e-4, "d image_feat")
    _close(T.grad, Tr.grad, 1e-4, "d text_feat")
    assert t.grad.shape == t.shape
    assert abs(float(t.grad) - float(tr.grad)) <= 1e-4 * max(abs(float(tr.grad)), 1e-3)


@pytest.mark.parametrize("G,H", [(14, 12), (24, 3), (30, 2), (3, 1), (33, 1)])
def test_relpos_grid_grad_matches_the_index_scatter(G, H):
    """Table gradient of the standard grid index (beit2.py:92-116) by the structured kernel against an fp64 index_add over the
    reference's relative_position_index; accumulates into dtable; same bits on every launch (one owner per entry, fixed order)."""
    Fx = _fx()
    from xfm_amd.beit2 import build_relative_position_index
    N, ld = G * G + 1, (G * G + 1 + 15) // 16 * 16
    index = build_relative_position_index(G, G).cuda()
    nrd = (2 * G - 1) ** 2 + 3
    dd = _rand((H, N, ld), 1.0, F32, seed=G)
    base = _rand((nrd, H), 1.0, F32, seed=G + 1)
    ref = base.double().clone()
    for h in range(H):
        ref[:, h].index_add_(0, index.reshape(-1), dd[h, :, :N].double().reshape(-1))
    outs = []
    for _ in range(2):
        dt = base.clone()
        Fx.relpos_grid_grad(dd, G, H, ld, dt)
        outs.append(dt)
    assert torch.equal(outs[0], outs[1])
    _close(outs[0], ref, 2e-6, "relpos grid gradient")


@pytest.mark.parametrize("N,E", [(32, 256), (7, 256), (300, 128)])
def test_itc_loss_with_image_ids_matches_the_soft_label_form(N, E):
    """Retrieval fine-tuning (xfm.py:705-713): captions that share an image id are each other's positives, labels = pos / pos.sum(1);
    loss, both feature gradients and the temperature gradient against fp64 torch."""
    from xfm_amd.ops import itc_loss
    F = torch.nn.functional
    I = F.normalize(_rand((N, E), 1.0, F32, seed=11), dim=-1).requires_grad_(True)
    T = F.normalize(_rand((N, E), 1.0, F32, seed=12) + 0.5 * I.detach(), dim=-1).requires_grad_(True)
    t = torch.tensor(0.07, device="cuda", requires_grad=True)
    idx = torch.tensor([(i * 7) % max(N // 3, 2) for i in range(N)], device="cuda")   # ids shared by ~3 rows each
    Ir, Tr, tr = (v.detach().double().requires_grad_(True) for v in (I, T, t))
    loss = itc_loss(I, T, t, idx=idx)
    logits = Ir @ Tr.t() / tr
    pos = torch.eq(idx.view(-1, 1), idx.view(1, -1)).double()
    labels = pos / pos.sum(1, keepdim=True)
    ref = (-torch.sum(F.log_softmax(logits, dim=1) * labels, dim=1).mean() - torch.sum(F.log_softmax(logits.t(), dim=1) * labels, dim=1).mean()) / 2
    (loss * 1.3).backward()
    (ref * 1.3).backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(abs(float(ref)), 1.0)
    _close(I.grad, Ir.grad, 1e-4, "d image_feat (idx)")
    _close(T.grad, Tr.grad, 1e-4, "d text_feat (idx)")
    assert abs(float(t.grad) - float(tr.grad)) <= 1e-4 * max(abs(float(tr.grad)), 1e-3)


def test_hard_negative_draws_with_image_ids_never_pick_the_same_image():
    """xfm.py:731-734: entries of the row's own image id carry weight 0; the others follow softmax + 1e-5 (chi-square as below)."""
    Fx = _fx()
    F = torch.nn.functional
    B, E, draws = 8, 256, 3000
    I = F.normalize(_rand((B, E), 1.0, F32, seed=5), dim=-1)
    T = F.normalize(_rand((B, E), 1.0, F32, seed=6) + 0.3 * I, dim=-1)
    temp = torch.tensor([0.25], device="cuda")
    idx = torch.tensor([5, 9, 5, 2, 7, 1, 9, 3], device="cuda")
    same = torch.eq(idx.view(-1, 1), idx.view(1, -1))
    sim = I @ T.t() / temp
    w_i2t = (F.softmax(sim, dim=1) + 1e-5).masked_fill(same, 0)
    w_t2i = (F.softmax(sim.t(), dim=1) + 1e-5).masked_fill(same, 0)
    cnt_t, cnt_i = torch.zeros(B, B, device="cuda"), torch.zeros(B, B, device="cuda")
    ar = torch.arange(B, device="cuda")
    for s in range(draws):
        im, tx = Fx.hard_negatives(I, T, temp, (77 << 32) | s, idx=idx)
        cnt_i[ar, im] += 1
        cnt_t[ar, tx] += 1
    for cnt, w in ((cnt_t, w_i2t), (cnt_i, w_t2i)):
        assert float(cnt[same].sum()) == 0, "a row drew an entry of its own image id"
        p = (w / w.sum(1, keepdim=True)).double().cpu()
        c = cnt.double().cpu()
        for r in range(B):
            keep = p[r] * draws >= 5
            exp = torch.cat([p[r][keep] * draws, (p[r][~keep].sum() * draws).reshape(1)])
            obs = torch.cat([c[r][keep], c[r][~keep].sum().reshape(1)])
            m = exp > 0
            assert float((((obs - exp) ** 2)[m] / exp[m]).sum()) < 40.0


def test_hard_negative_draws_follow_the_reference_weights():
    """Frequencies of 4000 draws per row against softmax(sim / temp) + 1e-5 with the own entry zeroed (xfm.py:727-744); never the
    own index; chi-square per row at p = 1e-4 (df = B - 2) with the 1e-5 floor entries pooled."""
    Fx = _fx()
    F = torch.nn.functional
    B, E, draws = 8, 256, 4000
    I = F.normalize(_rand((B, E), 1.0, F32, seed=3), dim=-1)
    T = F.normalize(_rand((B, E), 1.0, F32, seed=4) + 0.3 * I, dim=-1)
    temp = torch.tensor([0.25], device="cuda")
    sim = I @ T.t() / temp
    w_i2t = F.softmax(sim, dim=1) + 1e-5
    w_t2i = F.softmax(sim.t(), dim=1) + 1e-5
    w_i2t.fill_diagonal_(0)
    w_t2i.fill_diagonal_(0)
    cnt_t = torch.zeros(B, B, device="cuda")   # text negatives of image i
    cnt_i = torch.zeros(B, B, device="cuda")   # image negatives of text j
    ar = torch.arange(B, device="cuda")
    for s in range(draws):
        im, tx = Fx.hard_negatives(I, T, temp, (1234 << 32) | s)
        cnt_i[ar, im] += 1
        cnt_t[ar, tx] += 1
    for cnt, w in ((cnt_t, w_i2t), (cnt_i, w_t2i)):
        assert float(cnt.diagonal().sum()) == 0, "a row drew itself"
        p = (w / w.sum(1, keepdim=True)).double().cpu()
        c = cnt.double().cpu()
        for r in range(B):
            keep = p[r] * draws >= 5
            exp = torch.cat([p[r][keep] * draws, (p[r][~keep].sum() * draws).reshape(1)])
            obs = torch.cat([c[r][keep], c[r][~keep].sum().reshape(1)])
            m = exp > 0
            chi2 = float((((obs - exp) ** 2)[m] / exp[m]).sum())
            assert chi2 < 40.0, f"row {r}: chi2 = {chi2:.1f} over {int(m.sum())} cells"   # chi2(7) at 1e-4 is 29.9
    # same seed, same draw
    a = Fx.hard_negatives(I, T, temp, 99)
    b = Fx.hard_negatives(I, T, temp, 99)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("rows,D", [(24, 1536), (8, 3072), (8, 6144), (100, 768)])
@pytest.mark.parametrize("dtype", [BF16, F32])
def test_layernorm_gelu_head(rows, D, dtype):
    """GELU(LayerNorm(x)) in one kernel against torch (the Linear -> LayerNorm -> GELU heads, xfm.py:115-121); output bf16."""
    Fx = _fx()
    x = _rand((rows, D), 2.0, dtype, seed=1)
    w, b = _rand((D,), 1.0, F32, seed=2), _rand((D,), 0.5, F32, seed=3)
    dy = _rand((rows, D), 1.0, BF16, seed=4)
    y, mean, rstd = Fx.ln_fwd(x, w, b, 1e-5, gelu=True)
    xr = x.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5))
    ref.backward(dy.float())
    _close(y, ref, 6e-3, "ln+gelu fwd")
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dx = torch.empty((rows, D), dtype=dtype, device="cuda")
    Fx.ln_bwd(dy, x, mean, rstd, w, dg, db, gelu_b=b, **({"dx32": dx} if dtype == F32 else {"dx16": dx}))
    _close(dx, xr.grad, 1e-2, "ln+gelu dx")
    _close(dg, wr.grad, 5e-3, "ln+gelu dgamma")
    _close(db, br.grad, 5e-3, "ln+gelu dbeta")


@pytest.mark.parametrize("R,C", [(192, 2), (24, 1000), (5, 3)])
def test_small_ce_matches_cross_entropy(R, C):
    from xfm_amd.ops import small_ce
    x = _rand((R, C), 2.0, F32, seed=1).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    g = torch.Generator().manual_seed(5)
    labels = torch.randint(0, C, (R,), generator=g).cuda()
    loss = small_ce(x, labels)
    ref = torch.nn.functional.cross_entropy(xr, labels)
    (loss * 3).backward()
    (ref * 3).backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(abs(float(ref)), 1.0)
    _close(x.grad, xr.grad, 8e-3, "ce dlogits (stored as bf16)")


@pytest.mark.parametrize("M,N,K,nb", [(4388, 768, 768, 3), (1300, 768, 768, 2), (2200, 256, 384, 4), (500, 768, 768, 3), (900, 200, 768, 2)])
def test_gemm_tn_batch_equals_separate_calls(M, N, K, nb):
    """Several weight gradients of one shape in one launch (+ bias gradients on some of them) against one xfm_gemm_tn each; the last
    two shapes take the fallback (single split / N not a multiple of 128)."""
    Fx = _fx()
    dys = [_rand((M, N), 1.0, seed=10 + i) for i in range(nb)]
    xs = [_rand((M, K), 1.0, seed=20 + i) for i in range(nb)]
    dw_a = [torch.full((N, K), 0.5, dtype=F32, device="cuda") for _ in range(nb)]
    dw_b = [t.clone() for t in dw_a]
    db_a = [torch.zeros(N, dtype=F32, device="cuda") if i % 2 == 1 else None for i in range(nb)]
    db_b = [None if t is None else t.clone() for t in db_a]
    Fx.gemm_tn_batch(dys, xs, dw_a, db_a)
    for i in range(nb):
        Fx.gemm_tn(dys[i], xs[i], dw_b[i], dbias=db_b[i])
        ref = 0.5 + dys[i].float().t() @ xs[i].float()
        _close(dw_a[i], ref, 2e-3, f"batched dW[{i}]")
        _close(dw_a[i], dw_b[i], 1e-5, f"batched vs single dW[{i}]")
        if db_a[i] is not None:
            _close(db_a[i], dys[i].float().sum(0), 2e-3, f"batched dbias[{i}]")


@pytest.mark.parametrize("M,shapes", [
    (2048, [(768, 768), (1024, 256), (512, 512)]),                         # 17 tiles: every tile cut over the workgroups (stream-K only)
    (1536, [(1536, 768)] * 12 + [(3072, 768)] * 3),                        # 324 tiles: 256 whole ones, 68 cut over 102 workgroups
    (1024, [(768, 3072)] * 22),                                            # 792 tiles: three whole waves + 24 tiles, pieces = whole tiles
    (2048 + 40, [(768, 768), (320, 256), (256, 768), (768, 200)]),         # ragged M (tail rows) and shapes the 256 x 256 pipeline refuses
    (1152, [(256, 256)] * 50),                                             # more problems than one launch's table holds
    (1024, [(1024, 1024)] * 16),                                           # exactly one whole round of tiles: nothing is cut
    (1024, [(1024, 1024)] * 31),                                           # 496 tiles: the last round is 94 % full and runs whole too
    (1024, [(256, 768)]),                                                  # three tiles of the shortest allowed M: pieces = whole tiles
    (1024 + 63, [(512, 256), (256, 512)]),                                 # the last K-step holds 63 of its 64 rows
])
def test_gemm_tn_group_whole_tiles_and_stream_k_tail(M, shapes):
    """Grouped weight gradients (the deferred wgrads of a tower in persistent launches): values against fp32 math with dW += and the
    bias gradients of every other problem; exact on small-integer operands (a misplaced tile or piece cannot hide); and the same bits
    on every run -- whole tiles have one owner, the cut tiles' pieces are added in workgroup order."""
    Fx = _fx()
    n = len(shapes)
    dys = [_rand((M, N), 1.0, seed=100 + i) for i, (N, K) in enumerate(shapes)]
    xs = [_rand((M, K), 1.0, seed=200 + i) for i, (N, K) in enumerate(shapes)]

    def run(dys, xs, init):
        dws = [torch.full((N, K), init, dtype=F32, device="cuda") for N, K in shapes]
        dbs = [torch.full((N,), init, dtype=F32, device="cuda") if i % 2 == 0 else None for i, (N, K) in enumerate(shapes)]
        Fx.gemm_tn_group([(dys[i], xs[i], dws[i], dbs[i]) for i in range(n)])
        return dws, dbs

    dws, dbs = run(dys, xs, 0.5)
    for i in range(n):
        _close(dws[i], 0.5 + dys[i].float().t() @ xs[i].float(), 2e-4, f"grouped dW[{i}]")
        if dbs[i] is not None:
            _close(dbs[i], 0.5 + dys[i].float().sum(0), 2e-4, f"grouped dbias[{i}]")
    dws2, _ = run(dys, xs, 0.5)
    for i, (N, K) in enumerate(shapes):   # (the shapes that fall back to xfm_gemm_tn may take its atomics path)
        assert N % 256 or K % 256 or torch.equal(dws[i], dws2[i]), f"dW[{i}] differs between two runs"
    dyi = [((torch.arange(M * N, device="cuda").reshape(M, N) * (7 + i)) % 13 - 6).to(BF16) for i, (N, K) in enumerate(shapes)]
    xi = [((torch.arange(M * K, device="cuda").reshape(M, K) * (5 + i)) % 11 - 5).to(BF16) for i, (N, K) in enumerate(shapes)]
    dwi, dbi = run(dyi, xi, 0.0)
    for i in range(n):
        assert torch.equal(dwi[i], dyi[i].float().t() @ xi[i].float()), f"integer dW[{i}]"
        if dbi[i] is not None:
            assert torch.equal(dbi[i], dyi[i].float().sum(0)), f"integer dbias[{i}]"


@pytest.mark.parametrize("rows", [37, 5261, 25216])
def test_layernorm_backward_deferred_column_sums_equal_the_per_call_reduce(rows):
    """xfm_ln_bwd_args.defer + xfm_reduce_sets_batch (ONE fold for all the LayerNorms of a tower) against the reduce kernel that follows
    each backward kernel: same dx / dh, parameter gradients equal to the rounding of the final atomics; POST and layer-scale modes, several
    calls in one queue, gradients accumulated on top of existing values."""
    Fx = _fx()
    D = 768
    g = torch.Generator().manual_seed(rows)
    mk = lambda *shape, sc=1.0, dt=BF16: (torch.randn(*shape, generator=g) * sc).to(dt).cuda()   # noqa: E731
    w = mk(D, dt=F32) * 0.1 + 1.0
    outs = {}
    for mode in ("per call", "deferred"):
        rq = Fx.ReduceQueue(torch.device("cuda"), 4 * (Fx._lib.load().xfm_layernorm_bwd_workspace(rows, D, 2) + 256)) if mode == "deferred" else None
        torch.manual_seed(1)
        dg, db, dbias, dls = (torch.full((D,), 0.25, dtype=F32, device="cuda") for _ in range(4))
        res = []
        for call in range(2):   # two POST calls and two layer-scale calls into the same gradients
            gg = torch.Generator().manual_seed(100 + call)
            r = lambda *shape, sc=1.0, dt=BF16: (torch.randn(*shape, generator=gg) * sc).to(dt).cuda()   # noqa: E731
            dy, z = r(rows, D, sc=0.1), r(rows, D)
            mean, rstd = z.float().mean(1).contiguous(), (z.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
            res.append(Fx.ln_post_bwd(dy, z, mean, rstd, w, dg, db, dbias, drop=Fx.drop_params(0.1, 77 + call), defer=rq))
            x_new, h = r(rows, D, dt=F32), r(rows, D)
            dstream = r(rows, D, sc=0.1, dt=F32)
            mean2, rstd2 = x_new.mean(1).contiguous(), (x_new.var(1, unbiased=False) + 1e-6).rsqrt().contiguous()
            res.append((Fx.ln_ls_bwd(dy, dstream, x_new, mean2, rstd2, w, h, w, None, 1, dg, db, dbias, dls, defer=rq), dstream))
        if rq is not None:
            assert len(rq.items) == 4
            rq.run()
            rq.run()   # (a second run has nothing left to fold)
        outs[mode] = (res, (dg, db, dbias, dls))
    for a, b in zip(outs["per call"][0], outs["deferred"][0]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    for name, x, y in zip(("dgamma", "dbeta", "dbias", "dls"), outs["per call"][1], outs["deferred"][1]):
        _close(y, x, 2e-6, "deferred " + name)
        assert float((x - 0.25).abs().max()) > 1e-3, name   # (the reduce did add something)


@pytest.mark.parametrize("C", [2, 3, 5, 10, 101, 1000])
def test_small_ce_any_class_count_and_ignored_labels(C):
    """ops.small_ce = F.cross_entropy for every class count (the kernels read 4-column granules: widths that are not a multiple of 4 go
    through a padded copy) with ignore_index = -100 and the mean over the VALID rows, forward and backward."""
    from xfm_amd.ops import small_ce
    F = torch.nn.functional
    R = 37
    x = _rand((R, C), 2.0, F32, seed=C).requires_grad_(True)
    y = (torch.arange(R, device="cuda") * 7) % C
    y[::5] = -100
    xr = x.detach().double().requires_grad_(True)
    loss = small_ce(x, y)
    ref = F.cross_entropy(xr, y, ignore_index=-100)
    (loss * 1.3).backward()
    (ref * 1.3).backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(abs(float(ref)), 1.0)
    # dlogits leave the kernel as bf16
    _close(x.grad, xr.grad, 1e-2, "d logits")
    assert float(x.grad[::5].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,S,use_bias", [
    (5, 12, 197, True),     # the 224-px ViT: 7 key-tile pairs, Q / dO through LDS, query groups of 3 + 3 + 3 + 2 + 2 tiles
    (3, 2, 100, True),      # 4 pairs
    (2, 3, 250, True),      # 8 pairs: fragments stay in registers
    (3, 2, 40, False),      # waves without a key tile; no bias
])
def test_short_attention_backward_delta_from_output(B, H, S, use_bias, monkeypatch):
    """The short-sequence dQ kernel's opt-in form (XFM_ATTN_SHORT_PRE=1, csrc/attention.hip attn_bwd_dq_short_kernel<.., PRE = true>):
    the softmax-gradient row term delta_i = dO_i . (O_i + Olo_i) from the forward's output halves instead of the exchange of
    sum_j P_ij dP_ij between the key-range waves.  Held to the same tolerances against fp32 math as the default form, and to the
    default form's own results within bf16 noise; the returned delta to dO . O."""
    Fx = _fx()
    D = H * 64
    scale = 0.125
    qkv = _rand((B * S, 3 * D), seed=350 + S)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ld = (S + 15) // 16 * 16
    bias = None
    if use_bias:
        bias = torch.full((H, S, ld), float("nan"), dtype=F32, device="cuda")
        bias[:, :, :S] = _rand((H, S, S), 1.0, F32, seed=352)
    dout = _rand((B * S, D), seed=353)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    br = bias[:, :, :S].clone().requires_grad_(True) if use_bias else None
    _attn_ref(qr, kr, vr, B, H, S, S, scale, br).backward(dout.float())
    o, lse, o_lo = Fx.attn_fwd(q, k, v, B, H, S, S, scale, bias=bias, lo=True)
    got = {}
    for pre in ("0", "1"):
        monkeypatch.setenv("XFM_ATTN_SHORT_PRE", pre)   # (read per call)
        dqkv = torch.full((B * S, 3 * D), float("nan"), dtype=BF16, device="cuda")
        dbias = torch.zeros_like(bias).nan_to_num(0.0) if use_bias else None
        delta = Fx.attn_bwd(dout, q, k, v, o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, S, S, scale, bias=bias, dbias=dbias, o_lo=o_lo)
        torch.cuda.synchronize()
        got[pre] = (dqkv.float(), None if dbias is None else dbias.clone(), delta.clone())
        _close(dqkv[:, :D], qr.grad, 2e-2, "dq")
        _close(dqkv[:, D:2 * D], kr.grad, 2e-2, "dk")
        _close(dqkv[:, 2 * D:], vr.grad, 2e-2, "dv")
        if use_bias:
            _close(dbias[:, :, :S], br.grad, 2e-2, "dbias")
            assert float(dbias[:, :, S:].abs().max() if ld > S else 0.0) == 0.0
    want_delta = (dout.float() * (o.float() + o_lo.float())).view(B, S, H, 64).sum(-1).permute(0, 2, 1)
    _close(got["1"][2][:, :, :S], want_delta, 1e-4, "delta from the output halves")
    _close(got["1"][0], got["0"][0], 1e-2, "dqkv, the two forms")


@pytest.mark.parametrize("mode", ["0", "3", "4"])   # XFM_ATTN_VIT_BWD: 0 = split dQ + dK/dV kernels (default), 3 = single-pass kernel (13 tiles), 4 = single pass + the block-walking bias-gradient kernel
@pytest.mark.parametrize("tiled", [True, False])
@pytest.mark.parametrize("B,H,S,use_bias", [
    (2, 12, 197, True),     # the 224-px ViT (13 tiles: the seventh key-owner wave holds 5 keys, the last query pair is one tile)
    (41, 12, 197, True),    # 492 items, two per workgroup: a workgroup whose items straddle two heads flushes its bias gradient twice
    (3, 5, 197, False),     # models/vit.py: no relative-position bias
    (2, 3, 208, True),      # every tile full
    (4, 2, 100, True),      # 7 tiles: four key-owner waves, the rest idle
    (3, 4, 65, True),       # 5 tiles, one key past the last full tile
    (1, 1, 80, True),       # a single item
])
def test_vit_attention_fused_forward_backward(B, H, S, use_bias, tiled, mode, monkeypatch):
    """The batch-walking ViT kernels (csrc/attention_vit.hip: one workgroup per (batch entry, head) problem at a time; forward with the
    whole score row and the head's bias rows in registers; opt-in single-pass backward with S / dP computed once for dQ, dK, dV and the
    bias gradient) against fp32 math (beit2.py:126-166).  The single-pass backward defines delta_i = dO_i . O_i with the bf16 O."""
    if mode in ("3", "4") and ((use_bias and not tiled) or (S + 15) // 16 != 13):
        pytest.skip("the single-pass backward takes 13-tile sequences and the tiled bias only")
    monkeypatch.setenv("XFM_ATTN_VIT_BWD", mode)   # (the single-pass backward kernels are opt-in; the library reads the switch per call)
    Fx = _fx()
    D = H * 64
    scale = 0.125
    qkv = _rand((B * S, 3 * D), seed=150 + S)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ld = (S + 15) // 16 * 16
    bias = bias_t = None
    if use_bias:
        bias = torch.full((H, S, ld), float("nan"), dtype=F32, device="cuda")   # the padding of a bias row is never initialised
        bias[:, :, :S] = _rand((H, S, S), 1.0, F32, seed=152)
        bias_t = torch.full((H, S, ld), float("nan"), dtype=F32, device="cuda")
        bias_t[:, :, :S] = bias[:, :, :S].transpose(1, 2)
    dout = _rand((B * S, D), seed=153)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    br = bias[:, :, :S].clone().requires_grad_(True) if use_bias else None
    ref = _attn_ref(qr, kr, vr, B, H, S, S, scale, br)
    ref.backward(dout.float())
    if tiled and not use_bias:
        pytest.skip("no bias to tile")
    tiles = Fx.bias_tiles(bias, S, scale) if tiled else None   # accumulator-layout copies (one contiguous load per tile)
    o, lse = Fx.attn_fwd(q, k, v, B, H, S, S, scale, bias=bias, bias_tiles=tiles)
    _close(o, ref, 1e-2, "attention out")
    s_ref = (qr.detach().view(B, S, H, 64).permute(0, 2, 1, 3) @ kr.detach().view(B, S, H, 64).permute(0, 2, 3, 1)) * scale
    if use_bias:
        s_ref = s_ref + br.detach().unsqueeze(0)
    assert float((lse[:, :, :S] - s_ref.logsumexp(-1)).abs().max()) <= 2e-2
    dqkv = torch.full((B * S, 3 * D), float("nan"), dtype=BF16, device="cuda")
    dbias = torch.zeros_like(bias).nan_to_num(0.0) if use_bias else None
    delta = Fx.attn_bwd(dout, q, k, v, o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, S, S, scale, bias=bias, dbias=dbias,
                        bias_t=None if tiled else bias_t, bias_tiles=tiles)
    _close(dqkv[:, :D], qr.grad, 2e-2, "dq")
    _close(dqkv[:, D:2 * D], kr.grad, 2e-2, "dk")
    _close(dqkv[:, 2 * D:], vr.grad, 2e-2, "dv")
    if mode in ("3", "4"):   # (the split kernels' delta is the two-pass sum_j P_ij dP_ij)
        want_delta = (dout.float() * o.float()).view(B, S, H, 64).sum(-1).permute(0, 2, 1)
        _close(delta[:, :, :S], want_delta, 1e-3, "delta")
    if use_bias:
        _close(dbias[:, :, :S], br.grad, 2e-2, "dbias")
        assert float(dbias[:, :, S:].abs().max() if ld > S else 0.0) == 0.0
