"""HIP kernels through the C ABI against fp32 torch math on the same bf16-rounded inputs.

Tolerances: outputs are bf16 (8 significant bits) of fp32-accumulated sums, so the
bar is |err| <= 2^-7 * |ref| + a small absolute term scaled to the output's rms."""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from sfcvit import ops as o
    return o


def bf(t):
    return t.to(torch.bfloat16)


def close(got, ref, rel=1.0 / 128, abs_scale=1.0 / 64):
    got, ref = got.float(), ref.float()
    tol = rel * ref.abs() + abs_scale * ref.pow(2).mean().sqrt().clamp_min(1e-6)
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{int(bad.sum())}/{bad.numel()} off, max err {float((got - ref).abs().max())}, ref rms {float(ref.pow(2).mean().sqrt())}"


def gelu_ref(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


@pytest.mark.parametrize("akm", [False, True])
@pytest.mark.parametrize("bkm", [False, True])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 328), (1024, 576, 192), (40, 16, 1000)])
def test_gemm_layouts(ops, akm, bkm, M, N, K):
    if akm and M % 8:
        pytest.skip("k-major A needs M % 8 == 0")
    g = torch.Generator(device="cuda").manual_seed(1)
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    b = bf(torch.randn(N, K, device="cuda", generator=g))
    ref = a.float() @ b.float().t()
    am = a.t().contiguous() if akm else a
    bm = b.t().contiguous() if bkm else b
    c = ops.gemm(am, bm, a_kmajor=akm, b_kmajor=bkm)
    close(c, ref)
    c32 = ops.gemm(am, bm, a_kmajor=akm, b_kmajor=bkm, out_f32=True)
    assert c32.dtype == torch.float32
    close(c32, ref, rel=1e-3, abs_scale=1e-3)


@pytest.mark.parametrize("akm", [False, True])
@pytest.mark.parametrize("bkm", [False, True])
@pytest.mark.parametrize("M,N,K,splitk", [(512, 256, 128, 1), (256, 384, 192, 1), (768, 128, 64, 1), (1024, 768, 512, 1),
                                          (256, 256, 4096, 4), (512, 384, 2048, 3)])
def test_gemm_large_tile_lds_dma_kernel(ops, akm, bkm, M, N, K, splitk):
    # shapes eligible for the 256-wide LDS-DMA kernel (gemm256.hip), against fp32 math and
    # against the generic kernel on the same inputs
    g = torch.Generator(device="cuda").manual_seed(11)
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    b = bf(torch.randn(N, K, device="cuda", generator=g))
    ref = a.float() @ b.float().t()
    am = a.t().contiguous() if akm else a
    bm = b.t().contiguous() if bkm else b
    c = ops.gemm(am, bm, a_kmajor=akm, b_kmajor=bkm, splitk=splitk, force_generic=4)    # 4 = force the large-tile kernel
    close(c, ref)
    cg = ops.gemm(am, bm, a_kmajor=akm, b_kmajor=bkm, splitk=splitk, force_generic=True)
    assert torch.equal(c, cg)          # same k order, same fp32 chain: bit-identical


def test_gemm_large_tile_epilogues(ops):
    g = torch.Generator(device="cuda").manual_seed(12)
    M, N, K = 512, 384, 256
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K))
    bias = bf(torch.randn(N, device="cuda", generator=g))
    res = bf(torch.randn(M, N, device="cuda", generator=g))
    u = bf(torch.randn(M, N, device="cuda", generator=g))
    for kw in (dict(bias=bias), dict(bias=bias, act=ops.ACT_RELU), dict(bias=bias, residual=res),
               dict(bias=bias, act=ops.ACT_GELU), dict(aux_in=u, dact=ops.ACT_RELU), dict(aux_in=u, dact=ops.ACT_GELU),
               dict(residual=res, out_f32=True)):
        assert torch.equal(ops.gemm(a, w, force_generic=4, **kw), ops.gemm(a, w, force_generic=True, **kw)), kw
    y, pre = ops.gemm(a, w, bias=bias, act=ops.ACT_GELU, want_aux=True, force_generic=4)
    yg, preg = ops.gemm(a, w, bias=bias, act=ops.ACT_GELU, want_aux=True, force_generic=True)
    assert torch.equal(y, yg) and torch.equal(pre, preg)
    close(pre, a.float() @ w.float().t() + bias.float())


def test_gemm_asymmetric_identity(ops):
    # A = I with an asymmetric B catches a transposed C write (cdna_hip_programming.md §3)
    n = 128
    a = bf(torch.eye(n, device="cuda"))
    b = bf(torch.arange(n * n, device="cuda").reshape(n, n).float() % 251)
    assert torch.equal(ops.gemm(a, b).float(), b.float().t())
    assert torch.equal(ops.gemm(a, b, b_kmajor=True).float(), b.float())


def test_gemm_epilogues(ops):
    g = torch.Generator(device="cuda").manual_seed(2)
    M, N, K = 300, 264, 192
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K))
    bias = bf(torch.randn(N, device="cuda", generator=g))
    res = bf(torch.randn(M, N, device="cuda", generator=g))
    z = a.float() @ w.float().t() + bias.float()
    close(ops.gemm(a, w, bias=bias), z)
    close(ops.gemm(a, w, bias=bias, act=ops.ACT_RELU), torch.relu(z))
    close(ops.gemm(a, w, bias=bias, residual=res), z + res.float())
    y, pre = ops.gemm(a, w, bias=bias, act=ops.ACT_GELU, want_aux=True)
    close(pre, z)
    close(y, gelu_ref(z))
    # backward-style epilogues: mask by relu output / multiply by gelu'(pre)
    h = bf(torch.relu(torch.randn(M, N, device="cuda", generator=g)))
    close(ops.gemm(a, w, aux_in=h, dact=ops.ACT_RELU), (a.float() @ w.float().t()) * (h.float() > 0))
    u = bf(torch.randn(M, N, device="cuda", generator=g))
    uf = u.float().requires_grad_(True)
    gelu_ref(uf).sum().backward()
    close(ops.gemm(a, w, aux_in=u, dact=ops.ACT_GELU), (a.float() @ w.float().t()) * uf.grad)


@pytest.mark.parametrize("splitk", [2, 5, 16])
def test_gemm_splitk_weight_grad_shape(ops, splitk):
    g = torch.Generator(device="cuda").manual_seed(3)
    Mrows, Dout, Din = 3000, 192, 136          # dW[Dout, Din] = dY^T X, contraction over rows
    dy = bf(torch.randn(Mrows, Dout, device="cuda", generator=g))
    x = bf(torch.randn(Mrows, Din, device="cuda", generator=g))
    ref = dy.float().t() @ x.float()
    got = ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, splitk=splitk)
    close(got, ref)
    again = ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, splitk=splitk)
    assert torch.equal(got, again)            # slab reduction is order-fixed: bitwise reproducible
    close(ops.gemm(dy, x, a_kmajor=True, b_kmajor=True), ref)   # auto split


def test_gemm_rejects_bad_arguments(ops):
    from sfcvit._lib import SfcvitError
    a = bf(torch.randn(16, 12, device="cuda"))
    with pytest.raises(SfcvitError):
        ops.gemm(a, a)                          # K = 12 is not a multiple of 8
    with pytest.raises(SfcvitError):
        ops.gemm(a.cpu(), a.cpu())              # no CPU fallback


@pytest.mark.parametrize("M,D", [(4097, 768), (12544, 768), (4099, 1024)])
def test_layernorm_forward_two_rows_per_wave_is_the_one_row_kernel(ops, M, D):
    """From 4 096 rows on, D = 768 / 1 024 run the forward with two rows per wave (ln_fwd2_kernel): bit-identical output,
    mean and rstd to the one-row kernel -- same per-row arithmetic in the same order up to the lane a vector sits in --
    within one bf16 step at most where the wave sums associate differently; odd M (a last wave with one row)."""
    g = torch.Generator(device="cuda").manual_seed(9)
    x = bf(torch.randn(M, D, device="cuda", generator=g) * 2 + 0.5)
    gamma = bf(1 + 0.2 * torch.randn(D, device="cuda", generator=g))
    beta = bf(0.1 * torch.randn(D, device="cuda", generator=g))
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    y1, mean1, rstd1 = (torch.cat(t) for t in zip(*(ops.layernorm_fwd(x[i:i + 1024].contiguous(), gamma, beta) for i in range(0, M, 1024))))
    ref = torch.nn.functional.layer_norm(x.float(), (D,), gamma.float(), beta.float(), 1e-5)
    close(y, ref)
    assert torch.allclose(mean, mean1, rtol=0, atol=1e-5) and torch.allclose(rstd, rstd1, rtol=1e-5, atol=0)
    d = (y.float() - y1.float()).abs()
    assert (d <= 2.0 ** -7 * y1.float().abs() + 1e-6).all()
    assert float((d > 0).float().mean()) <= 1e-3


@pytest.mark.parametrize("M,D", [(7, 192), (1000, 768), (513, 1024), (64, 128)])
def test_layernorm(ops, M, D):
    g = torch.Generator(device="cuda").manual_seed(4)
    x = bf(torch.randn(M, D, device="cuda", generator=g) * 2 + 0.5)
    gamma = bf(1 + 0.2 * torch.randn(D, device="cuda", generator=g))
    beta = bf(0.1 * torch.randn(D, device="cuda", generator=g))
    dy = bf(torch.randn(M, D, device="cuda", generator=g))
    add = bf(torch.randn(M, D, device="cuda", generator=g))
    xf = x.float().requires_grad_(True)
    gf, bfl = gamma.float().requires_grad_(True), beta.float().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xf, (D,), gf, bfl, 1e-5)
    ref.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    close(y, ref.detach())
    assert torch.allclose(mean, x.float().mean(-1), atol=1e-4)
    dx, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, gamma)
    close(dx, xf.grad)
    close(dg, gf.grad, rel=2e-3, abs_scale=2e-3)
    close(db, bfl.grad, rel=2e-3, abs_scale=2e-3)
    dx2, _, _ = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dx_add=add)
    close(dx2, xf.grad + add.float())
    dx3, dg3, db3, dcol = ops.layernorm_bwd(dy, x, mean, rstd, gamma, want_colsum=True)
    assert torch.equal(dx3, dx) and torch.equal(dg3, dg)
    close(dcol, dx.float().sum(0), rel=1e-2, abs_scale=1e-2)          # bias gradient of the sub-layer (fp32 sums of the un-rounded dx)


def test_colsum(ops):
    x = bf(torch.randn(5000, 328, device="cuda"))
    close(ops.colsum(x), x.float().sum(0), rel=1e-3, abs_scale=1e-3)
    v = x[:, 64:192]                               # strided view of a packed buffer
    close(ops.colsum(v), v.float().sum(0), rel=1e-3, abs_scale=1e-3)


def attn_ref(qkv, H):
    B, N, D3 = qkv.shape
    D = D3 // 3
    hd = D // H
    q, k, v = qkv.float().split(D, dim=-1)
    sp = lambda t: t.reshape(B, N, H, hd).transpose(1, 2)
    s = (sp(q) @ sp(k).transpose(-1, -2)) / math.sqrt(hd)
    p = torch.softmax(s, -1)
    o = (p @ sp(v)).transpose(1, 2).reshape(B, N, D)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,N,H", [(2, 4, 3), (3, 196, 2), (2, 64, 1), (1, 576, 2), (2, 130, 4)])
def test_attention(ops, B, N, H):
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = bf(torch.randn(B, N, 3 * H * 64, device="cuda", generator=g))
    dout = bf(torch.randn(B, N, H * 64, device="cuda", generator=g))
    qf = qkv.float().requires_grad_(True)
    ref, lse_ref = attn_ref(qf, H)
    ref.backward(dout.float())
    out, lse = ops.attention_fwd(qkv, H)
    close(out, ref.detach())
    assert torch.allclose(lse, lse_ref.detach(), atol=2e-2, rtol=1e-2)
    dqkv = ops.attention_bwd(qkv, out, lse, dout, H)
    close(dqkv, qf.grad, rel=1.0 / 64, abs_scale=1.0 / 32)


@pytest.mark.parametrize("B,N,H", [(4, 196, 2), (2, 64, 1), (1, 576, 1)])
def test_attention_is_run_to_run_deterministic(ops, B, N, H):
    # regression: hipcc missed the MFMA -> VALU read hazard across a taken branch (the row max read
    # accumulators before the MFMA had written them: correct to rounding, different every run)
    g = torch.Generator(device="cuda").manual_seed(9)
    qkv = bf(torch.randn(B, N, 3 * H * 64, device="cuda", generator=g))
    dout = bf(torch.randn(B, N, H * 64, device="cuda", generator=g))
    o0, l0 = ops.attention_fwd(qkv, H)
    d0 = ops.attention_bwd(qkv, o0, l0, dout, H)
    for _ in range(10):
        o, l = ops.attention_fwd(qkv, H)
        assert torch.equal(o, o0) and torch.equal(l, l0)
        assert torch.equal(ops.attention_bwd(qkv, o0, l0, dout, H), d0)
    q, k, _ = qkv.float().split(H * 64, dim=-1)
    sp = lambda t: t.reshape(B, N, H, 64).transpose(1, 2)
    true_lse = torch.logsumexp(sp(q) @ sp(k).transpose(-1, -2) / 8.0, -1)
    assert torch.allclose(l0, true_lse, atol=2e-2, rtol=1e-2)


def test_attention_spiked_row(ops):
    # one key dominating one query forces the running-max rescale across key blocks
    B, N, H = 1, 196, 1
    g = torch.Generator(device="cuda").manual_seed(6)
    qkv = torch.randn(B, N, 3 * 64, device="cuda", generator=g)
    qkv[0, 5, :64] *= 6
    qkv[0, 150, 64:128] = qkv[0, 5, :64]          # key 150 (third block) aligned with query 5
    qkv = bf(qkv)
    ref, _ = attn_ref(qkv, H)
    out, _ = ops.attention_fwd(qkv, H)
    close(out, ref)


def test_soft_ce(ops):
    g = torch.Generator(device="cuda").manual_seed(7)
    B, C, ld = 37, 10, 16
    logits = torch.zeros(B, ld, device="cuda")
    logits[:, :C] = torch.randn(B, C, device="cuda", generator=g) * 3
    logits = bf(logits)
    t = torch.softmax(torch.randn(B, C, device="cuda", generator=g), -1)
    lf = logits[:, :C].float().requires_grad_(True)
    loss = -(t * torch.log_softmax(lf, -1)).sum(-1)
    loss.mean().backward()
    rows, dl = ops.soft_ce(logits, t.contiguous(), C, 1.0 / B)
    assert torch.allclose(rows, loss.detach(), atol=1e-4, rtol=1e-4)
    close(dl[:, :C], lf.grad)
    assert not dl[:, C:].any()


def test_gelu(ops):
    x = bf(torch.randn(4096, device="cuda") * 2)
    dy = bf(torch.randn(4096, device="cuda"))
    xf = x.float().requires_grad_(True)
    y = gelu_ref(xf)
    y.backward(dy.float())
    close(ops.gelu_fwd(x), y.detach())
    close(ops.gelu_bwd(dy, x), xf.grad)


def test_adamw_and_clip(ops):
    g = torch.Generator(device="cuda").manual_seed(8)
    n = 10007
    w0 = torch.randn(n, device="cuda", generator=g)
    ref = torch.nn.Parameter(w0.clone())
    opt = torch.optim.AdamW([ref], lr=3e-4, weight_decay=5e-5)
    master, m, v = w0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    param = bf(w0)
    for step in range(1, 4):
        grad = bf(torch.randn(n, device="cuda", generator=g) * 0.1)
        ref.grad = grad.float()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        ss = torch.zeros(1, device="cuda")
        ops.sumsq_accum(grad, ss)
        assert torch.allclose(ss, grad.float().pow(2).sum(), rtol=1e-4)
        ops.adamw_step(param, master, grad, m, v, ss, lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-8,
                       weight_decay=5e-5, max_norm=1.0, step=step)
        assert torch.allclose(master, ref.detach(), atol=1e-6, rtol=1e-5)
        assert torch.equal(param, bf(master))


def test_transpose_and_gemm_dx(ops):
    g = torch.Generator(device="cuda").manual_seed(21)
    w = bf(torch.randn(2304, 768, device="cuda", generator=g))
    assert torch.equal(ops.transpose(w), w.t().contiguous())
    v = w[:, 128:392]                                   # strided view, 264 columns
    assert torch.equal(ops.transpose(v), v.t().contiguous())
    dy = bf(torch.randn(512, 2304, device="cuda", generator=g))
    res = bf(torch.randn(512, 768, device="cuda", generator=g))
    ref = dy.float() @ w.float() + res.float()
    close(ops.gemm_dx(dy, w, residual=res), ref)                         # LDS-DMA kernel on W^T
    close(ops.gemm_dx(dy[:200], w, residual=res[:200]), ref[:200])       # generic kernel, W read k-major


def test_gemm_row_slices_share_one_dropout_mask(ops):
    # a GEMM computed as two row slices (row_offset) reproduces the single-launch dropout mask
    g = torch.Generator(device="cuda").manual_seed(31)
    M, N, K = 1024, 256, 64
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g))
    kw = dict(act=ops.ACT_RELU, dropout_p=0.1, dropout_seed=99)
    full = ops.gemm(a, w, **kw)
    lo = ops.gemm(a[:512], w, **kw)
    hi = ops.gemm(a[512:], w, row_offset=512, **kw)
    assert torch.equal(full, torch.cat([lo, hi]))
    assert torch.equal(full, ops.gemm(a, w, force_generic=1, **kw))


# ---------------------------------------------------------------------------------------------------
# persistent 8-phase kernel (gemm8p.hip): force_generic 8 = 256-row tiles, 9 = 224-row tiles
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,mode", [
    (256, 256, 256, 8),          # one tile, four k-tiles (the shortest k loop the kernel takes)
    (512, 512, 256, 8),          # 4 tiles
    (224, 256, 256, 9),          # one 224-row tile
    (448, 768, 384, 9),
    (256 * 37, 256 * 7, 256, 8),     # 259 tiles for 256 workgroups: some draw a third tile, short k loop
    (224 * 40, 256 * 7, 256, 9),     # 280 tiles
    (256 * 12, 768, 768, 8),         # ViT-B k loop
    (192, 256, 256, 10),             # one 192-row tile
    (192 * 41, 1024, 1024, 10),      # ViT-L out-projection shape, 164 tiles
])
def test_gemm_persistent_kernel_against_fp32_and_generic(ops, M, N, K, mode):
    g = torch.Generator(device="cuda").manual_seed(21)
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) * 0.1)
    bias = bf(torch.randn(N, device="cuda", generator=g))
    c = ops.gemm(a, w, bias=bias, force_generic=mode)
    close(c, a.float() @ w.float().t() + bias.float())
    assert torch.equal(c, ops.gemm(a, w, bias=bias, force_generic=1))
    assert torch.equal(ops.gemm(a, w, force_generic=mode), ops.gemm(a, w, force_generic=1))


@pytest.mark.parametrize("mode,M", [(8, 1024), (9, 896), (10, 768)])
def test_gemm_persistent_kernel_epilogues(ops, mode, M):
    """Every epilogue variant the persistent kernel is built for gives bit-identical results to the generic kernel
    (same fp32 accumulation order per k-tile is not guaranteed in general, so first compare against fp32 math)."""
    N, K = 512, 256
    g = torch.Generator(device="cuda").manual_seed(22)
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) * 0.1)
    bias = bf(torch.randn(N, device="cuda", generator=g))
    res = bf(torch.randn(M, N, device="cuda", generator=g))
    aux = bf(torch.randn(M, N, device="cuda", generator=g))
    variants = [dict(bias=bias, residual=res), dict(bias=bias, residual=res, dropout_p=0.1, dropout_seed=77),
                dict(bias=bias, act=ops.ACT_RELU), dict(bias=bias, act=ops.ACT_RELU, dropout_p=0.1, dropout_seed=5),
                dict(aux_in=aux, dact=ops.ACT_RELU, dact_scale=1.0 / 0.9), dict(residual=res), dict()]
    for kw in variants:
        got, want = ops.gemm(a, w, force_generic=mode, **kw), ops.gemm(a, w, force_generic=1, **kw)
        assert (got.float() - want.float()).abs().max() <= 2.0 ** -6 * want.float().abs().max(), kw
        same = (got == want).float().mean().item()
        assert same > 0.98, (kw, same)                      # identical up to rare 1-ulp bf16 rounding flips
    # ReLU + residual is not one of the built variants: automatic dispatch falls back, forcing it is an error
    y = ops.gemm(a, w, bias=bias, act=ops.ACT_RELU, residual=res)
    close(y, torch.relu(a.float() @ w.float().t() + bias.float()) + res.float())
    with pytest.raises(Exception, match="not eligible"):
        ops.gemm(a, w, bias=bias, act=ops.ACT_RELU, residual=res, force_generic=mode)


@pytest.mark.parametrize("mode,M", [(8, 1000), (9, 1001), (10, 200), (0, 19600), (0, 49000)])
def test_gemm_persistent_kernel_ragged_rows(ops, mode, M):
    """M that no tile height divides (ViT-B at batch 100 / 250: 19 600 / 49 000 token rows; a last partial batch): the
    persistent kernel's last row tile overlaps the one before it.  Every epilogue variant against the generic kernel
    as in test_gemm_persistent_kernel_epilogues; the rows are bit-identical to the same rows of a single launch on a
    tile-aligned problem (the ragged GEMM extended by extra rows), dropout mask included; fused column sums count the
    shared rows once."""
    N, K = (768, 768) if mode == 0 else (512, 256)
    g = torch.Generator(device="cuda").manual_seed(24)
    tile = {8: 256, 9: 224, 10: 192, 0: 256}[mode]
    Mp = (M + tile - 1) // tile * tile                                 # the aligned problem the ragged one is a prefix of
    a = bf(torch.randn(Mp, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) * 0.1)
    bias = bf(torch.randn(N, device="cuda", generator=g))
    res = bf(torch.randn(Mp, N, device="cuda", generator=g))
    aux = bf(torch.randn(Mp, N, device="cuda", generator=g))
    variants = [dict(bias=bias, residual=res), dict(bias=bias, residual=res, dropout_p=0.1, dropout_seed=77),
                dict(bias=bias, act=ops.ACT_RELU, dropout_p=0.1, dropout_seed=5),
                dict(aux_in=aux, dact=ops.ACT_RELU, dact_scale=1.0 / 0.9), dict()]
    for kw in variants:
        cut = {k: (v[:M] if k in ("residual", "aux_in") else v) for k, v in kw.items()}
        ops.KERNEL_LOG = []
        got = ops.gemm(a[:M], w, force_generic=mode, **cut)
        ran, ops.KERNEL_LOG = ops.KERNEL_LOG, None
        assert ran[0].startswith("gemm8p_kernel<"), ran
        want = ops.gemm(a[:M], w, force_generic=1, **cut)
        assert (got.float() - want.float()).abs().max() <= 2.0 ** -6 * want.float().abs().max(), kw
        assert (got == want).float().mean().item() > 0.98, kw
        full = ops.gemm(a, w, force_generic=mode if mode else 8, **kw)          # aligned: no overlapping tile
        assert torch.equal(got, full[:M]), kw
    # fused column sums (DACT | CSUM) over a ragged M: exact up to the usual rounding, bit-reproducible
    kw = dict(aux_in=aux[:M], dact=ops.ACT_RELU, dact_scale=1.0 / 0.9)
    c, cs = ops.gemm(a[:M], w, colsum=True, force_generic=mode, **kw)
    exact = ((a[:M].float() @ w.float().t()) * (aux[:M].float() > 0) / 0.9).sum(0)
    tol = 4.0 * math.sqrt(M) * 2.0 ** -9 * c.float().pow(2).mean().sqrt()
    assert (cs - exact).abs().max() <= tol
    assert torch.equal(cs, ops.gemm(a[:M], w, colsum=True, force_generic=mode, **kw)[1])
    # the activation bit mask written / read over a ragged M
    bits = torch.zeros((M, N // 8), device="cuda", dtype=torch.uint8)
    h = ops.gemm(a[:M], w, bias=bias, act=ops.ACT_RELU, actmask=bits, force_generic=mode)
    packed = ((h.float() > 0).view(M, N // 8, 8).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(-1)
    assert torch.equal(bits, packed.to(torch.uint8))


@pytest.mark.parametrize("M,N,K", [(768, 768, 19600), (256, 512, 16384 + 72), (768, 2304, 49000)])
def test_gemm_weight_gradient_ragged_k(ops, M, N, K):
    """dW = dY^T X when the token rows are not a multiple of 128 (batch 100: k = 19 600): the 8-phase weight-gradient
    kernel on the largest multiple of 128 + one more slab for the rest, against fp32 math, bit-reproducible."""
    g = torch.Generator(device="cuda").manual_seed(32)
    a = bf(torch.randn(K, M, device="cuda", generator=g))
    b = bf(torch.randn(K, N, device="cuda", generator=g))
    ops.KERNEL_LOG = []
    c = ops.gemm(a, b, a_kmajor=True, b_kmajor=True)
    ran, ops.KERNEL_LOG = ops.KERNEL_LOG, None
    assert ran == ["gemm8p_km_kernel<true>"], ran
    close(c, a.float().t() @ b.float())
    assert torch.equal(c, ops.gemm(a, b, a_kmajor=True, b_kmajor=True))
    # the tail rows matter: zeroing them changes the (fp32) result by exactly their contribution
    a0 = a.clone()
    a0[K // 128 * 128:] = 0
    c32 = ops.gemm(a, b, a_kmajor=True, b_kmajor=True, out_f32=True)
    c0 = ops.gemm(a0, b, a_kmajor=True, b_kmajor=True, out_f32=True)
    tail = a[K // 128 * 128:].float().t() @ b[K // 128 * 128:].float()
    close(c32 - c0, tail, rel=1 / 64, abs_scale=1 / 64)


def test_gemm_persistent_kernel_is_the_automatic_choice_and_deterministic(ops):
    M, N, K = 256 * 9, 768, 768
    g = torch.Generator(device="cuda").manual_seed(23)
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) * 0.1)
    first = ops.gemm(a, w)
    assert torch.equal(first, ops.gemm(a, w, force_generic=8))
    for _ in range(5):
        assert torch.equal(first, ops.gemm(a, w))


@pytest.mark.parametrize("M,N,K,splitk", [(256, 256, 4096, 4), (512, 768, 8192, 8), (768, 768, 50176, None),
                                          (768, 2304, 12544, None), (256, 512, 640, 2), (256, 256, 1024 + 128, 3)])
def test_gemm_weight_gradient_8phase_kernel(ops, M, N, K, splitk):
    """dW = dY^T X with both operands k-major: the split-K 8-phase kernel (automatic choice for these shapes)
    against fp32 math, and run-to-run bit-identical (ordered slab reduction, no atomics)."""
    g = torch.Generator(device="cuda").manual_seed(31)
    a = bf(torch.randn(K, M, device="cuda", generator=g))
    b = bf(torch.randn(K, N, device="cuda", generator=g))
    ref = a.float().t() @ b.float()
    c = ops.gemm(a, b, a_kmajor=True, b_kmajor=True, splitk=splitk)
    close(c, ref)
    c32 = ops.gemm(a, b, a_kmajor=True, b_kmajor=True, splitk=splitk, out_f32=True)
    close(c32, ref, rel=2e-3, abs_scale=2e-3)
    assert torch.equal(c, ops.gemm(a, b, a_kmajor=True, b_kmajor=True, splitk=splitk))
    if splitk is not None:
        assert torch.equal(c, ops.gemm(a, b, a_kmajor=True, b_kmajor=True, splitk=splitk, force_generic=8))
    # asymmetric check: a one-hot A picks single rows of B (catches a transposed or permuted fragment map)
    a1 = torch.zeros(K, M, device="cuda", dtype=torch.bfloat16)
    idx = torch.randint(0, K, (M,), device="cuda", generator=g)
    a1[idx, torch.arange(M, device="cuda")] = 1
    want = torch.zeros(M, N, device="cuda")
    want.index_add_(0, torch.arange(M, device="cuda"), b.float()[idx])
    assert torch.equal(ops.gemm(a1, b, a_kmajor=True, b_kmajor=True, splitk=splitk).float(), want.bfloat16().float())


@pytest.mark.parametrize("B,N,H,hd", [(2, 64, 4, 192), (2, 196, 2, 128), (1, 100, 1, 256), (3, 180, 1, 192), (2, 5, 2, 128)])
def test_attention_wide_heads(ops, B, N, H, hd):
    """Head dims 128 / 192 / 256 (the reference's own script uses 768 / 4 = 192, main.py:276-282): forward, lse and the
    packed dqkv against fp32 math; run-to-run identical."""
    g = torch.Generator(device="cuda").manual_seed(15)
    qkv = bf(torch.randn(B, N, 3 * H * hd, device="cuda", generator=g))
    dout = bf(torch.randn(B, N, H * hd, device="cuda", generator=g))
    qf = qkv.float().requires_grad_(True)
    ref, lse_ref = attn_ref(qf, H)
    ref.backward(dout.float())
    out, lse = ops.attention_fwd(qkv, H)
    close(out, ref.detach())
    assert torch.allclose(lse, lse_ref.detach(), atol=2e-2, rtol=1e-2)
    dqkv = ops.attention_bwd(qkv, out, lse, dout, H)
    close(dqkv, qf.grad, rel=1.0 / 64, abs_scale=1.0 / 32)
    o2, l2 = ops.attention_fwd(qkv, H)
    assert torch.equal(o2, out) and torch.equal(l2, lse) and torch.equal(ops.attention_bwd(qkv, out, lse, dout, H), dqkv)


def test_attention_wide_heads_dropout_and_limits(ops):
    B, N, H, hd, p, seed = 2, 64, 4, 192, 0.1, 99
    D = H * hd
    g = torch.Generator(device="cuda").manual_seed(16)
    qkv = bf(torch.randn(B, N, 3 * D, device="cuda", generator=g))
    dout = bf(torch.randn(B, N, D, device="cuda", generator=g))
    mask = ops.dropout_mask(B * H * N, N, p, seed).float().view(B, H, N, N)
    qf = qkv.float().requires_grad_(True)
    q, k, v = qf.split(D, dim=-1)
    sp = lambda t: t.reshape(B, N, H, hd).transpose(1, 2)
    s = (sp(q) @ sp(k).transpose(-1, -2)) / math.sqrt(hd)
    ref = ((torch.softmax(s, -1) * mask) @ sp(v)).transpose(1, 2).reshape(B, N, D)
    ref.backward(dout.float())
    out, lse = ops.attention_fwd(qkv, H, p, seed)
    close(out, ref.detach())
    close(ops.attention_bwd(qkv, out, lse, dout, H, p, seed), qf.grad, rel=1 / 48, abs_scale=1 / 24)
    with pytest.raises(Exception, match="head dim"):            # 96 is not a multiple of 64
        ops.attention_fwd(bf(torch.randn(1, 16, 3 * 96, device="cuda")), 1)
    with pytest.raises(Exception, match="LDS"):                 # 576 tokens x head dim 128 do not fit one CU's LDS
        ops.attention_fwd(bf(torch.randn(1, 576, 3 * 128, device="cuda")), 1)


@pytest.mark.parametrize("D,H,p", [(128, 4, 0.0), (192, 4, 0.0), (96, 1, 0.0), (128, 4, 0.1)])
def test_encoder_layer_with_head_dims_the_kernels_do_not_have(D, H, p):
    """embed_dim / n_heads = 32, 48, 96: constructor-compatible reference models (nn.TransformerEncoderLayer takes any
    divisor) run on the next head dim the attention kernels have, through zero-padded in_proj rows / out_proj columns and
    the true 1 / sqrt(head dim) softmax scale.  Against torch's own layer in fp32 on the same bf16-rounded weights:
    output and every parameter gradient (dropout 0); with dropout only finiteness and run-to-run mask freshness."""
    from sfcvit import functional as F
    torch.manual_seed(3)
    B, N, Fd = 3, 50, 2 * D
    ref = torch.nn.TransformerEncoderLayer(D, H, Fd, dropout=0.0, batch_first=True).cuda()
    with torch.no_grad():
        for q in ref.parameters():
            q.copy_(q.bfloat16().float())
    names = ["self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
             "norm1.weight", "norm1.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
             "norm2.weight", "norm2.bias"]
    sd = dict(ref.named_parameters())
    mine = [sd[n].detach().clone().requires_grad_(True) for n in names]
    x = bf(torch.randn(B, N, D, device="cuda")).float()
    r = torch.randn(B, N, D, device="cuda")
    xr = x.clone().requires_grad_(True)
    (ref(xr) * r).sum().backward()
    xm = x.clone().requires_grad_(True)
    y = F.encoder_layer(xm, *mine, H, dropout_p=p)
    assert y.shape == (B, N, D) and torch.isfinite(y.float()).all()
    if p > 0:
        assert not torch.equal(y, F.encoder_layer(xm, *mine, H, dropout_p=p))
        return
    close(y, ref(x).detach(), rel=1 / 64, abs_scale=1 / 32)
    (y.float() * r).sum().backward()
    for n, t, tr in [("x", xm, xr)] + [(n, t, sd[n]) for n, t in zip(names, mine)]:
        g, gr = t.grad.flatten(), tr.grad.flatten()
        cos = float(torch.dot(g, gr) / (g.norm() * gr.norm() + 1e-30))
        assert cos >= 0.995, (n, cos)
        assert abs(float(g.norm() / gr.norm()) - 1) <= 3e-2, n


def test_attention_op_with_padded_head_dim():
    """functional.attention on a packed projection with head dim 48: zero-padded to 64 inside, true softmax scale."""
    from sfcvit import functional as F
    g = torch.Generator(device="cuda").manual_seed(8)
    B, N, H, hd = 2, 70, 3, 48
    qkv = bf(torch.randn(B, N, 3 * H * hd, device="cuda", generator=g))
    dout = torch.randn(B, N, H * hd, device="cuda", generator=g)
    qf = qkv.float().requires_grad_(True)
    q, k, v = qf.split(H * hd, dim=-1)
    sp = lambda t: t.reshape(B, N, H, hd).transpose(1, 2)
    ref = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(hd), -1) @ sp(v)).transpose(1, 2).reshape(B, N, H * hd)
    ref.backward(dout)
    qm = qkv.clone().requires_grad_(True)
    out = F.attention(qm, H)
    close(out, ref.detach())
    out.backward(bf(dout))
    close(qm.grad, qf.grad, rel=1 / 48, abs_scale=1 / 24)


@pytest.mark.parametrize("M,N,K,p", [(896, 512, 256, 0.1), (1792, 768, 256, 0.0), (448, 256, 384, 0.1), (200, 144, 64, 0.1)])
def test_gemm_activation_bit_mask(ops, M, N, K, p):
    """sfcvit_gemm_args.actmask: the sign pattern of a ReLU (+ dropout) output as a bit matrix, written by the forward GEMM
    (fused into the persistent kernel's epilogue for the first three shapes, a pass over C for the last) and read by the
    dX GEMM in place of the activation itself -- bits == (C > 0) exactly, C unchanged by asking for them, and the
    gradient GEMM bit-identical with and without the mask (with and without fused column sums)."""
    g = torch.Generator(device="cuda").manual_seed(17)
    x = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K))
    b = bf(torch.randn(N, device="cuda", generator=g))
    kw = dict(bias=b, act=ops.ACT_RELU, dropout_p=p, dropout_seed=41)
    h0 = ops.gemm(x, w, **kw)
    bits = torch.full((M, N // 8 + 2), 0xAB, device="cuda", dtype=torch.uint8)[:, : N // 8 + 2]
    h = ops.gemm(x, w, actmask=bits, **kw)
    assert torch.equal(h, h0)
    want = (h.float() > 0).view(M, N // 8, 8).to(torch.int32)
    packed = (want << torch.arange(8, device="cuda", dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits[:, : N // 8], packed)
    assert (bits[:, N // 8:] == 0xAB).all()                          # nothing written beyond N / 8 bytes per row
    dy = bf(torch.randn(M, K, device="cuda", generator=g))
    w2 = bf(torch.randn(K, N, device="cuda", generator=g) / math.sqrt(K))      # linear2: [out = K, in = N]
    for cs in (None, True):
        ref = ops.gemm_dx(dy, w2, aux_in=h, dact=ops.ACT_RELU, dact_scale=1.0 / (1.0 - p), colsum=cs)
        got = ops.gemm_dx(dy, w2, aux_in=h, dact=ops.ACT_RELU, dact_scale=1.0 / (1.0 - p), colsum=cs, actmask=bits)
        if cs:
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        else:
            assert torch.equal(got, ref)
    with pytest.raises(Exception, match="actmask"):
        ops.gemm(x, w, bias=b, actmask=bits)                          # neither act nor dact is RELU


@pytest.mark.parametrize("M,N,K", [(896, 512, 256), (1024, 768, 384), (200, 136, 64)])
def test_gemm_fused_column_sums(ops, M, N, K):
    """colsum= : column sums of the epilogue's result next to C (bias gradient of the previous Linear).  Fused into
    the persistent kernel's epilogue for the activation-gradient variant (first two shapes), a separate pass otherwise
    (third shape: off the tile grid) -- same C either way, sums within fp32-vs-bf16 rounding of each other."""
    g = torch.Generator(device="cuda").manual_seed(41)
    a = bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) * 0.1)
    aux = bf(torch.randn(M, N, device="cuda", generator=g))
    kw = dict(aux_in=aux, dact=ops.ACT_RELU, dact_scale=1.0 / 0.9)
    c0 = ops.gemm(a, w, **kw)
    c, cs = ops.gemm(a, w, colsum=True, **kw)
    assert torch.equal(c, c0) and cs.dtype == torch.float32
    # the fused path sums the fp32 values before their bf16 rounding (closer to the exact sum), the separate pass sums
    # the stored bf16 C: both within sqrt(M) * bf16 rounding of the exact fp32 column sums
    exact = ((a.float() @ w.float().t()) * (aux.float() > 0) / 0.9).sum(0)
    tol = 4.0 * math.sqrt(M) * 2.0 ** -9 * c0.float().pow(2).mean().sqrt()
    assert (cs - exact).abs().max() <= tol
    assert (cs - c0.float().sum(0)).abs().max() <= tol
    slot = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    c2, cs2 = ops.gemm(a, w, colsum=slot, **kw)
    assert cs2.data_ptr() == slot.data_ptr() and torch.equal(c2, c0)
    close(slot, exact)
    for _ in range(3):                                  # fixed-order reduction: bit-reproducible
        assert torch.equal(ops.gemm(a, w, colsum=True, **kw)[1], cs)
    # a variant without a fused form (bias + colsum) takes the separate pass
    bias = bf(torch.randn(N, device="cuda", generator=g))
    c3, cs3 = ops.gemm(a, w, bias=bias, colsum=True)
    close(cs3, c3.float().sum(0), rel=2e-3, abs_scale=2e-3)


def test_ops_are_hipgraph_capturable(ops):
    """The reference wraps its model in torch.compile(mode="reduce-overhead") (main.py:284), i.e. HIP graphs: nothing in
    the C ABI may allocate, synchronise or touch the host at launch time.  Capture a GEMM on the persistent kernel (its
    tile-queue counters come from a pre-allocated pool), a LayerNorm and an attention forward on a fresh capture
    stream, replay on new inputs."""
    g = torch.Generator(device="cuda").manual_seed(51)
    a = bf(torch.randn(512, 256, device="cuda", generator=g))
    w = bf(torch.randn(512, 256, device="cuda", generator=g) * 0.1)
    bias = bf(torch.randn(512, device="cuda", generator=g))
    gam, bet = bf(torch.ones(512, device="cuda")), bf(torch.zeros(512, device="cuda"))
    qkv = bf(torch.randn(2, 64, 3 * 128, device="cuda", generator=g))
    ops.gemm(a, w, bias=bias)                       # warm-up outside the capture (first call creates the pool)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = ops.gemm(a, w, bias=bias)
        z, _, _ = ops.layernorm_fwd(y, gam, bet, 1e-5)
        o, _ = ops.attention_fwd(qkv, 2)
    a2 = bf(torch.randn(512, 256, device="cuda", generator=g))
    q2 = bf(torch.randn(2, 64, 3 * 128, device="cuda", generator=g))
    a.copy_(a2)
    qkv.copy_(q2)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ops.gemm(a2, w, bias=bias))
    assert torch.equal(z, ops.layernorm_fwd(y, gam, bet, 1e-5)[0])
    assert torch.equal(o, ops.attention_fwd(q2, 2)[0])


@pytest.mark.parametrize("B,N,H,p", [(3, 196, 2, 0.0), (2, 196, 3, 0.1), (2, 224, 1, 0.1), (2, 33, 2, 0.0), (1, 200, 2, 0.1),
                                     (2, 4, 1, 0.1), (1, 209, 1, 0.0)])
def test_attention_bwd_fused_vs_two_kernel_form(ops, B, N, H, p, monkeypatch):
    """The single-pass backward (attention_bwd_fused.hip: dK, dV and dQ from one evaluation of P and dS, dS handed to
    the dQ waves through LDS) against the two-kernel form (SFCVIT_ATTN_BWD_FUSED=0) and against fp32 math; run twice
    for bit-reproducibility (every dQ tile is summed by one wave in a fixed order)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    D, seed = H * 64, 777
    qkv = bf(torch.randn(B, N, 3 * D, device="cuda", generator=g))
    dout = bf(torch.randn(B, N, D, device="cuda", generator=g))
    out, lse = ops.attention_fwd(qkv, H, p, seed)
    monkeypatch.setenv("SFCVIT_ATTN_BWD_FUSED", "0")
    two = ops.attention_bwd(qkv, out, lse, dout, H, p, seed)
    monkeypatch.setenv("SFCVIT_ATTN_BWD_FUSED", "1")
    one = ops.attention_bwd(qkv, out, lse, dout, H, p, seed)
    again = ops.attention_bwd(qkv, out, lse, dout, H, p, seed)
    assert torch.equal(one, again)
    close(one, two.float(), rel=1 / 128, abs_scale=1 / 64)
    mask = ops.dropout_mask(B * H * N, N, p, seed).float().view(B, H, N, N) if p > 0 else 1.0
    qf = qkv.float().requires_grad_(True)
    q, k, v = qf.split(D, dim=-1)
    sp = lambda t: t.reshape(B, N, H, 64).transpose(1, 2)
    s = (sp(q) @ sp(k).transpose(-1, -2)) / 8.0
    ref = ((torch.softmax(s, -1) * mask) @ sp(v)).transpose(1, 2).reshape(B, N, D)
    ref.backward(dout.float())
    close(one, qf.grad, rel=1 / 48, abs_scale=1 / 24)


@pytest.mark.parametrize("curve,img,D,B,xdt", [("hilbert", 224, 768, 3, torch.float32), ("hilbert", 224, 256, 130, torch.float32),
                                               ("z", 384, 1024, 2, torch.float32), ("raster", 224, 768, 2, torch.bfloat16),
                                               ("hilbert", 32, 256, 37, torch.float32), ("z", 32, 512, 4, torch.bfloat16)])
def test_patch_embed_tiled_vs_generic_and_reference(ops, curve, img, D, B, xdt):
    """The tiled gather + projection kernel (csrc/patch_embed_tiled.hip: 16-pixel row segments, class-permuted weight)
    against the generic pixel-table kernel and against plain torch indexing with the SAME table: identical products,
    another summation order (fp32 accumulate), so bf16-rounding-sized differences only.  Covers ragged class sizes
    (rows padded to whole 128-row tiles), 1 / 3 / 4 classes, strips, fp32 and bf16 images."""
    from sfcvit.tokenizers.embeddings import _pixel_table
    from sfcvit.curves import curve_table, hilbert_curve, z_curve
    g = torch.Generator(device="cuda").manual_seed(21)
    flat = np.arange(img * img, dtype=np.int32) if curve == "raster" else curve_table({"hilbert": hilbert_curve, "z": z_curve}[curve], img)
    pix_h = _pixel_table(flat, img, 1, 256)
    pix = torch.from_numpy(pix_h).cuda()
    desc = ops.tile_descriptor(pix_h, img, "cuda")
    assert desc is not None
    x = torch.randn(B, 3, img, img, device="cuda", generator=g).to(xdt)
    w = bf(torch.randn(D, 768, device="cuda", generator=g) / 28)
    b = bf(torch.randn(D, device="cuda", generator=g))
    tiled = ops.patch_embed_fwd(x, pix, w, b, desc)
    generic = ops.patch_embed_fwd(x.to(torch.bfloat16), pix, w, b, None)
    close(tiled, generic.float(), rel=1 / 128, abs_scale=1 / 128)
    tok = x.to(torch.bfloat16).float().reshape(B, 3, img * img)[:, :, pix.long()]       # [B, C, N, P]
    tok = tok.permute(0, 2, 3, 1).reshape(B, pix.shape[0], 768)                           # feature = k*C + c
    ref = tok @ w.float().t() + b.float()
    close(tiled, ref, rel=1 / 100, abs_scale=1 / 100)
    assert torch.equal(tiled, ops.patch_embed_fwd(x, pix, w, b, desc))


@pytest.mark.parametrize("B,N,H,p", [(2, 576, 2, 0.0), (1, 576, 3, 0.1), (2, 577, 1, 0.1), (1, 257, 2, 0.0), (2, 300, 1, 0.1),
                                     (1, 600, 1, 0.0), (1, 608, 2, 0.1)])
def test_attention_long_kernels_vs_tiled_and_reference(ops, B, N, H, p, monkeypatch):
    """The sequence-resident kernels for 256 < N <= 608 (attention_long.hip: 12 waves per (batch, head); forward with
    192-key softmax chunks, dK/dV with Q and dO resident, dQ with K and V resident) against the tiled kernels
    (SFCVIT_ATTN_LONG=0) with the SAME dropout mask and against fp32 math with that mask; log-sum-exp against torch;
    bit-reproducible.  Covers N = 576 (the compile-time forward instance), ragged N (257, 577), padding fragments
    (300, 600) and the limit 608."""
    g = torch.Generator(device="cuda").manual_seed(31)
    D, seed = H * 64, 4242
    qkv = torch.randn(B, N, 3 * D, device="cuda", generator=g)
    qkv[0, 7, :64] *= 5                                   # one peaked row: the running max moves between key chunks
    qkv[0, N - 3, D:D + 64] = qkv[0, 7, :64]
    qkv = bf(qkv)
    dout = bf(torch.randn(B, N, D, device="cuda", generator=g))
    plain = bf(torch.randn(B, N, 3 * D, device="cuda", generator=g))
    monkeypatch.setenv("SFCVIT_ATTN_LONG", "0")
    o_t, l_t = ops.attention_fwd(qkv, H, p, seed)
    o_pt, l_pt = ops.attention_fwd(plain, H, p, seed)
    d_t = ops.attention_bwd(plain, o_pt, l_pt, dout, H, p, seed)
    monkeypatch.setenv("SFCVIT_ATTN_LONG", "1")
    o_l, l_l = ops.attention_fwd(qkv, H, p, seed)
    o_2, l_2 = ops.attention_fwd(qkv, H, p, seed)
    assert torch.equal(o_l, o_2) and torch.equal(l_l, l_2)
    close(o_l, o_t.float())
    assert torch.allclose(l_l, l_t, atol=2e-3, rtol=1e-4)
    mask = ops.dropout_mask(B * H * N, N, p, seed).float().view(B, H, N, N) if p > 0 else 1.0
    sp = lambda t: t.reshape(B, N, H, 64).transpose(1, 2)

    def fp32(x):
        xf = x.float().requires_grad_(True)
        q, k, v = xf.split(D, dim=-1)
        s = (sp(q) @ sp(k).transpose(-1, -2)) / 8.0
        ref = ((torch.softmax(s, -1) * mask) @ sp(v)).transpose(1, 2).reshape(B, N, D)
        ref.backward(dout.float())
        return ref.detach(), torch.logsumexp(s.detach(), -1), xf.grad

    ref, lse_ref, _ = fp32(qkv)
    close(o_l, ref)
    assert torch.allclose(l_l, lse_ref, atol=2e-2, rtol=1e-2)
    # Backward on the un-spiked input: where P -> 1 the flash form dS = P (dP - delta) cancels (delta comes from the
    # bf16-rounded O) and bf16-rounding-sized differences in O are amplified, in the tiled kernels as much as here.
    o_p, l_p = ops.attention_fwd(plain, H, p, seed)
    d_l = ops.attention_bwd(plain, o_p, l_p, dout, H, p, seed)
    assert torch.equal(d_l, ops.attention_bwd(plain, o_p, l_p, dout, H, p, seed))
    close(d_l, d_t.float(), rel=1 / 64, abs_scale=1 / 32)      # both round P and dS to bf16, in different association
    close(d_l, fp32(plain)[2], rel=1 / 64, abs_scale=1 / 32)


@pytest.mark.parametrize("B,N,H,p", [(3, 196, 2, 0.1), (2, 100, 3, 0.0), (2, 300, 1, 0.0), (2, 576, 2, 0.1), (1, 601, 1, 0.1), (2, 700, 1, 0.0)])
def test_attention_bwd_column_sums(ops, B, N, H, p):
    """The in_proj bias gradient as a by-product of the attention backward: 192 sums per (batch, head) out of the one-pass
    kernel (N <= 224), out of the sequence-resident dQ and dK/dV kernels (256 < N <= 608) or from a column-sum pass over
    dqkv (other lengths: 300 -> long, 700 -> tiled + pass) -- against the column sums of the dqkv the same
    call returned (fp32 sums of the bf16-rounded values differ from the kernel's sums of the unrounded ones by rounding)."""
    g = torch.Generator(device="cuda").manual_seed(13)
    D = H * 64
    qkv = bf(torch.randn(B, N, 3 * D, device="cuda", generator=g))
    dout = bf(torch.randn(B, N, D, device="cuda", generator=g))
    out, lse = ops.attention_fwd(qkv, H, p, 99)
    plain = ops.attention_bwd(qkv, out, lse, dout, H, p, 99)
    dqkv, cs = ops.attention_bwd(qkv, out, lse, dout, H, p, 99, colsum=True)
    assert torch.equal(dqkv, plain)
    ref = dqkv.float().sum((0, 1))
    close(cs, ref, rel=1 / 100, abs_scale=1 / 100)
    slot = torch.zeros(3 * D, device="cuda", dtype=torch.bfloat16)
    _, cs2 = ops.attention_bwd(qkv, out, lse, dout, H, p, 99, colsum=slot)
    assert cs2 is slot and torch.equal(slot, cs.to(torch.bfloat16))


@pytest.mark.parametrize("curve,img,D,B,xdt", [("hilbert", 224, 768, 3, torch.float32), ("hilbert", 224, 256, 70, torch.bfloat16),
                                               ("z", 384, 1024, 2, torch.float32), ("raster", 224, 768, 2, torch.float32),
                                               ("hilbert", 32, 256, 37, torch.float32)])
def test_patch_embed_tiled_backward(ops, curve, img, D, B, xdt):
    """dW and dbias of the tiled gather kernels (row segments straight into k-major LDS images, per-class partial
    gradients un-permuted by the reduction) against the generic pixel-table kernels and fp32 torch math."""
    from sfcvit.tokenizers.embeddings import _pixel_table
    from sfcvit.curves import curve_table, hilbert_curve, z_curve
    g = torch.Generator(device="cuda").manual_seed(22)
    flat = np.arange(img * img, dtype=np.int32) if curve == "raster" else curve_table({"hilbert": hilbert_curve, "z": z_curve}[curve], img)
    pix_h = _pixel_table(flat, img, 1, 256)
    pix = torch.from_numpy(pix_h).cuda()
    desc = ops.tile_descriptor(pix_h, img, "cuda")
    N = pix.shape[0]
    x = torch.randn(B, 3, img, img, device="cuda", generator=g).to(xdt)
    dy = bf(torch.randn(B, N, D, device="cuda", generator=g))
    dw, db = ops.patch_embed_bwd(x, pix, dy, D, True, desc)
    dw_g, db_g = ops.patch_embed_bwd(x.to(torch.bfloat16), pix, dy, D, True, None)
    close(dw, dw_g, rel=1 / 256, abs_scale=1 / 256)
    close(db, db_g, rel=1e-4, abs_scale=1e-4)
    tok = x.to(torch.bfloat16).float().reshape(B, 3, img * img)[:, :, pix.long()].permute(0, 2, 3, 1).reshape(B * N, 768)
    ref = dy.float().reshape(B * N, D).t() @ tok
    close(dw, ref, rel=1 / 200, abs_scale=1 / 200)
    dw2, _ = ops.patch_embed_bwd(x, pix, dy, D, True, desc)
    assert torch.equal(dw, dw2)


def test_transpose_batched_matches_torch(ops):
    """sfcvit_transpose_batched: matrices of mixed shapes (edges that are not multiples of the 64 x 64 tile) packed in one
    flat buffer, one launch, against torch's .t()."""
    import numpy as np
    shapes = [(768, 256), (64, 64), (136, 72), (8, 200), (256, 1032)]
    g = torch.Generator(device="cuda").manual_seed(2)
    mats = [torch.randn(r, c, device="cuda", generator=g).to(torch.bfloat16) for r, c in shapes]
    offs, off = [], 0
    for m in mats:
        offs.append(off)
        off += (m.numel() + 7) // 8 * 8
    src = torch.zeros(off, device="cuda", dtype=torch.bfloat16)
    for m, o in zip(mats, offs):
        src[o:o + m.numel()] = m.flatten()
    dst = torch.full((off,), -1.0, device="cuda", dtype=torch.bfloat16)
    rows = [(o, o, r, c, r0, c0) for (r, c), o in zip(shapes, offs) for r0 in range(0, r, 64) for c0 in range(0, c, 64)]
    table = np.array(rows, dtype=np.dtype([("src_off", "<i8"), ("dst_off", "<i8"), ("R", "<i4"), ("C", "<i4"), ("r0", "<i4"), ("c0", "<i4")]))
    tiles = torch.from_numpy(table.view(np.uint8).copy()).cuda()
    ops.transpose_batched(src, dst, tiles, len(rows))
    for m, o in zip(mats, offs):
        assert torch.equal(dst[o:o + m.numel()].view(m.shape[1], m.shape[0]), m.t().contiguous())


@pytest.mark.parametrize("curve,img,P,C,B,xdt", [("hilbert", 224, 256, 3, 3, torch.float32), ("z", 384, 256, 3, 2, torch.float32),
                                                 ("raster", 224, 256, 3, 9, torch.bfloat16), ("hilbert", 32, 16, 3, 37, torch.float32),
                                                 ("hilbert", 32, 64, 1, 5, torch.float32), ("z", 32, 4, 3, 2, torch.float32)])
def test_tokens_gather_is_the_reference_gather(ops, curve, img, P, C, B, xdt):
    """sfcvit_tokens_gather against torch indexing with the same pixel table (hilbert_embedding1D.py:36-41: x_flat[:, :, perm]
    -> [B, N, P * C], feature = k * C + c): bit-exact (one bf16 rounding of the same pixels); rows padded to 8 columns are
    zero-filled (P * C = 12 here)."""
    from sfcvit.tokenizers.embeddings import _pixel_table
    from sfcvit.curves import curve_table, hilbert_curve, z_curve
    g = torch.Generator(device="cuda").manual_seed(3)
    flat = np.arange(img * img, dtype=np.int32) if curve == "raster" else curve_table({"hilbert": hilbert_curve, "z": z_curve}[curve], img)
    pix = torch.from_numpy(_pixel_table(flat, img, 1, P)).cuda()
    x = torch.randn(B, C, img, img, device="cuda", generator=g).to(xdt)
    tokens = ops.gather_tokens(x, pix)
    N, K = pix.shape[0], P * C
    ref = x.to(torch.bfloat16).reshape(B, C, img * img)[:, :, pix.long()].permute(0, 2, 3, 1).reshape(B * N, K)
    assert tokens.shape == (B * N, (K + 7) // 8 * 8)
    assert torch.equal(tokens[:, :K], ref)
    assert not tokens[:, K:].any()


@pytest.mark.parametrize("curve,img,C,B", [("hilbert", 224, 3, 3), ("z", 384, 3, 2), ("hilbert", 32, 1, 37), ("z", 64, 4, 9),
                                           ("hilbert", 224, 3, 64), ("hilbert", 48, 2, 8), ("hilbert", 32, 3, 5), ("z", 48, 3, 7)])
def test_tokens_gather_tiles_is_the_reference_gather(ops, curve, img, C, B):
    """sfcvit_tokens_gather_tiles (16 x 16 tile tokens of an fp32 image: whole image lines in, curve order applied out of
    LDS) against torch indexing with the same pixel table and against the per-pixel kernel: bit-exact; with and without
    the order (with it and 3 channels: the strip kernel, one workgroup per row of tiles; otherwise tile pairs); batches
    that are not a multiple of a workgroup's images; 1-4 channels; an odd number of tiles per row (48 px: 3)."""
    from sfcvit.tokenizers.embeddings import _pixel_table
    from sfcvit.curves import curve_table, hilbert_curve, z_curve
    g = torch.Generator(device="cuda").manual_seed(5)
    flat = curve_table({"hilbert": hilbert_curve, "z": z_curve}[curve], img)
    pix_h = _pixel_table(flat, img, 1, 256)
    desc = ops.tile_descriptor(pix_h, img, "cuda")
    assert desc is not None and desc.mode == 1
    pix = torch.from_numpy(pix_h).cuda()
    order = torch.from_numpy(ops.gather_order(pix_h)).cuda()
    x = torch.randn(B, C, img, img, device="cuda", generator=g)
    N, K = pix.shape[0], 256 * C
    ref = x.to(torch.bfloat16).reshape(B, C, img * img)[:, :, pix.long()].permute(0, 2, 3, 1).reshape(B * N, K)
    assert ops.GATHER_TILES
    for od in (order, None):
        tokens = ops.gather_tokens(x, pix, desc, od)
        assert tokens.shape == (B * N, K)
        assert torch.equal(tokens, ref), (curve, img, C, B, od is None)
    ops.GATHER_TILES = False
    try:
        assert torch.equal(ops.gather_tokens(x, pix, desc, order), ref)
    finally:
        ops.GATHER_TILES = True


@pytest.mark.parametrize("counts,D,B", [((256, 64), 32, 2), ((64, 64, 16), 64, 3), ((64, 256), 40, 2), ((100, 36, 9, 100), 8, 5), ((7, 1), 16, 1)])
def test_hier_resample_concat_is_torch_linear_interpolation(ops, counts, D, B):
    """sfcvit_hier_resample_concat / _bwd against F.interpolate(mode="linear", align_corners=False) + cat on the same bf16
    level outputs (multi_hilbert.py:33-38): forward within one bf16 step of the fp32 result (up-, down- and identity
    resampling, non-integer ratios, a single-token level); backward (the transposed resampling, fixed order) against
    autograd of the torch form in fp32: 1e-2 of the largest gradient (bf16 outputs)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    levels = [torch.randn(B, n, D, device="cuda", generator=g).bfloat16() for n in counts]
    out = ops.hier_resample_concat(levels)
    leaves = [t.float().requires_grad_(True) for t in levels]
    parts = [leaves[0]] + [t if t.shape[1] == counts[0] else
                           torch.nn.functional.interpolate(t.transpose(1, 2), size=counts[0], mode="linear", align_corners=False).transpose(1, 2)
                           for t in leaves[1:]]
    ref = torch.cat(parts, dim=-1)
    assert out.shape == ref.shape
    assert ((out.float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 1e-6).all()
    dout = torch.randn(ref.shape, device="cuda", generator=g).bfloat16()
    ref.backward(dout.float())
    grads = ops.hier_resample_concat_bwd(dout, list(counts), D)
    for got, leaf in zip(grads, leaves):
        assert got.shape == leaf.shape
        assert (got.float() - leaf.grad).abs().max() <= 1e-2 * leaf.grad.abs().max() + 1e-6


@pytest.mark.parametrize("M,N,K,akm,bkm", [(12544, 768, 768, False, False), (12544, 3072, 768, False, False), (4096, 768, 3072, False, False),
                                           (768, 2304, 12544, True, True), (1000, 264, 200, False, False), (392, 768, 768, False, True)])
def test_gemm_results_are_the_correctly_rounded_fp32_sums(ops, M, N, K, akm, bkm):
    """Bit level, not tolerance: every GEMM family (persistent forward / dX, the k-major weight-gradient kernel with its fp32
    split-K slabs, the older kernels for odd shapes) stores the bf16 rounding of the fp32 sum -- equal to rounding torch's
    fp32 product of the same bf16 operands except where the two summation orders' last bits straddle a rounding boundary
    (then adjacent bf16 values): <= 1e-3 of the elements, none further than one step."""
    g = torch.Generator(device="cuda").manual_seed(17)
    a = bf(torch.randn((K, M) if akm else (M, K), device="cuda", generator=g))
    b = bf(torch.randn((K, N) if bkm else (N, K), device="cuda", generator=g) * K ** -0.5)
    bias = bf(torch.randn(N, device="cuda", generator=g) * 0.1) if not (akm or bkm) else None
    got = ops.gemm(a, b, a_kmajor=akm, b_kmajor=bkm, bias=bias).float()
    ref = (a.float().t() if akm else a.float()) @ (b.float() if bkm else b.float().t())
    if bias is not None:
        ref = ref + bias.float()
    ref = ref.to(torch.bfloat16).float()
    d = (got - ref).abs()
    assert (d <= 2.0 ** -7 * torch.maximum(got.abs(), ref.abs()) + 1e-5 * ref.abs().max()).all(), float(d.max())
    assert float((d > 0).float().mean()) <= 1e-3, (ops.last_gemm_kernel(), float((d > 0).float().mean()))


@pytest.mark.parametrize("B,H,N,p", [(3, 4, 196, 0.1), (2, 2, 196, 0.0), (2, 3, 64, 0.1), (1, 2, 576, 0.1)])
def test_attention_backward_against_fp32_math_with_the_kernel_roundings(ops, B, H, N, p):
    """dQ | dK | dV of the attention backward (one-pass kernel for N <= 224, sequence-resident kernels above) against fp32
    math that rounds where the kernels round -- P (masked, scaled) and dS to bf16 before their MFMA products, fp32 sums, one
    bf16 rounding of each result -- from the forward's own lse and the same dropout mask.  Each output is two roundings deep
    (like the forward's, tests/test_parity_gpu.py), so the allowance on top of one bf16 step is every P / dS of the sum one
    step off: 2^-8 sum |terms| (used: <= 0.35 of it); at most 5e-3 of the elements differ at all (measured: 4e-5 .. 1.7e-3)."""
    hd, D = 64, H * 64
    g = torch.Generator(device="cuda").manual_seed(23)
    qkv = bf(torch.randn(B, N, 3 * D, device="cuda", generator=g))
    dout = bf(torch.randn(B, N, D, device="cuda", generator=g))
    seed = 77
    out, lse = ops.attention_fwd(qkv, H, p, seed)
    dqkv = ops.attention_bwd(qkv, out, lse, dout, H, p, seed).float()
    rb = lambda t: t.to(torch.bfloat16).float()                                   # noqa: E731
    sp = lambda t: t.reshape(B, N, H, hd).transpose(1, 2)                          # noqa: E731
    q, k, v = (sp(t) for t in qkv.float().split(D, dim=-1))
    do, o = sp(dout.float()), sp(out.float())
    scale = hd ** -0.5
    ks = 1.0 / (1.0 - p)
    keep = (ops.dropout_mask(B * H * N, N, p, seed).float().view(B, H, N, N) > 0).float() * ks if p > 0 else torch.ones(B, H, N, N, device="cuda")
    P = torch.exp((q @ k.transpose(-1, -2)) * scale - lse.unsqueeze(-1))           # lse: the forward's, natural log
    delta = (do * o).sum(-1, keepdim=True)
    Pm = rb(P * keep)
    dS = rb(P * ((do @ v.transpose(-1, -2)) * keep - delta))
    ref = {"dq": (dS @ k) * scale, "dk": (dS.transpose(-1, -2) @ q) * scale, "dv": Pm.transpose(-1, -2) @ do}
    slack = {"dq": (dS.abs() @ k.abs()) * scale, "dk": (dS.abs().transpose(-1, -2) @ q.abs()) * scale, "dv": Pm.transpose(-1, -2) @ do.abs()}
    unsp = lambda t: t.transpose(1, 2).reshape(B, N, D)                            # noqa: E731
    for i, name in enumerate(("dq", "dk", "dv")):
        got, want = dqkv[..., i * D:(i + 1) * D], rb(unsp(ref[name]))
        d = (got - want).abs()
        bound = 2.0 ** -7 * torch.maximum(got.abs(), want.abs()) + 2.0 ** -8 * unsp(slack[name]) + 1e-6
        assert (d <= bound).all(), (name, ops.last_attn_kernel(), float((d / bound).max()))
        assert float((d > 0).float().mean()) <= 5e-3, (name, ops.last_attn_kernel(), float((d > 0).float().mean()))
        print(f"[attention backward, bit level] N={N} p={p} {name}: {float((d > 0).float().mean()):.2e} of the elements differ, worst {float((d / bound).max()):.2f} of the bound [{ops.last_attn_kernel()}]")


@pytest.mark.parametrize("M,D", [(4097, 768), (1000, 1024), (333, 192)])
def test_rowwise_kernels_store_the_correctly_rounded_value(ops, M, D):
    """LayerNorm forward / backward (dx, with and without the added residual gradient) and GELU forward / backward at bit
    level: the bf16 rounding of the fp32 formula on the same bf16 inputs, up to one bf16 step where the fp32 evaluation
    orders differ (reductions, erf / exp implementations): <= 2e-3 of the elements."""
    g = torch.Generator(device="cuda").manual_seed(29)
    x = bf(torch.randn(M, D, device="cuda", generator=g) * 1.5 + 0.3)
    gamma, beta = bf(1 + 0.2 * torch.randn(D, device="cuda", generator=g)), bf(0.1 * torch.randn(D, device="cuda", generator=g))
    dy, add = bf(torch.randn(M, D, device="cuda", generator=g)), bf(torch.randn(M, D, device="cuda", generator=g))

    def same(tag, got, ref, frac=2e-3):
        got, ref = got.float(), ref.to(torch.bfloat16).float()
        d = (got - ref).abs()
        assert (d <= 2.0 ** -7 * torch.maximum(got.abs(), ref.abs()) + 1e-5 * ref.abs().max()).all(), (tag, float(d.max()))
        assert float((d > 0).float().mean()) <= frac, (tag, float((d > 0).float().mean()))

    xf = x.float().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xf, (D,), gamma.float(), beta.float(), 1e-5)
    ref.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    same("layernorm forward", y, ref.detach())
    same("layernorm backward dx", ops.layernorm_bwd(dy, x, mean, rstd, gamma)[0], xf.grad)
    same("layernorm backward dx + add", ops.layernorm_bwd(dy, x, mean, rstd, gamma, dx_add=add)[0], xf.grad + add.float())
    uf = x.float().requires_grad_(True)
    gr = torch.nn.functional.gelu(uf)
    gr.backward(dy.float())
    same("gelu forward", ops.gelu_fwd(x), gr.detach())
    same("gelu backward", ops.gelu_bwd(dy, x), uf.grad)
