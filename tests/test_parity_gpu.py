"""End-to-end parity of the HIP path (through the C ABI) with the oracle and with the
fixtures generated from the reference itself.

Stated tolerance (north star: "matching logits within a stated fp tolerance"): the HIP
path computes in bf16 with fp32 accumulation, the oracle / reference in fp32, so
   max|logit_hip - logit_ref| <= 3e-2 * max|logit_ref|            (eval mode)
   |loss_hip - loss_ref|      <= 2e-3 * |loss_ref| + 2e-3
   per-parameter gradient: cosine >= 0.99 and |‖g_hip‖/‖g_ref‖ - 1| <= 5e-2 (bf16 grads)
Curve index buffers must be identical (bit-exact)."""
import json
import math
import os

import pytest
import torch

from oracle import formula, vit_oracle
from oracle.cases import MODEL_CASES
from test_host_cpu import build_model

pytestmark = pytest.mark.gpu

LOGIT_TOL = 3e-2
R_LOGIT_TOL, R_COS = 2e-2, 0.995      # end to end against the rounding-point oracle (see the store-point test for why not tighter)


def load_formula(model, cfg):
    sd = vit_oracle.formula_state(cfg)
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return sd


@pytest.mark.parametrize("param_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", sorted(MODEL_CASES))
def test_logits_loss_and_grads(name, param_dtype, golden_dir):
    import sfcvit.functional as F
    cfg, batch = MODEL_CASES[name]
    with open(os.path.join(golden_dir, f"model_{name}.json")) as f:
        gold = json.load(f)
    model = build_model(cfg)
    sd = load_formula(model, cfg)
    if cfg.table_key:
        assert torch.equal(model.state_dict()["patch_embed." + cfg.table_key], sd["patch_embed." + cfg.table_key])
    model = model.to("cuda", dtype=param_dtype).eval()
    x = formula.image_batch(batch, cfg.in_channels, cfg.img_size, cfg.img_size)
    tgt = formula.soft_targets(batch, cfg.num_classes)

    # oracle (fp32 CPU) on the same state
    leaves = {k: v.requires_grad_(True) for k, v in vit_oracle.trainable(sd).items()}
    ref_logits, inter = vit_oracle.forward(x, sd, cfg, return_intermediates=True)
    ref_loss = vit_oracle.soft_target_ce(ref_logits, tgt)
    ref_loss.backward()

    tokens = model.patch_embed(x.cuda())
    tok_ref = inter["tokens"].detach()
    assert (tokens.float().cpu() - tok_ref).abs().max() <= 2e-2 * tok_ref.abs().max()

    logits = model(x.cuda())
    assert logits.shape == (batch, cfg.num_classes)
    got = logits.float().cpu()
    scale = ref_logits.detach().abs().max()
    assert (got - ref_logits.detach()).abs().max() <= LOGIT_TOL * scale, (got, ref_logits)
    # and against the reference's own output (fixture)
    gold_logits = torch.tensor(gold["logits"], dtype=torch.float32)
    assert (got - gold_logits).abs().max() <= LOGIT_TOL * gold_logits.abs().max()

    loss = F.soft_target_cross_entropy(logits, tgt.cuda())
    assert abs(float(loss.detach()) - gold["loss"]) <= 2e-3 * abs(gold["loss"]) + 2e-3
    loss.backward()
    for k, p in model.named_parameters():
        if k.startswith("mlp_mixer.token_mix"):
            assert p.grad is None, k            # unused in the reference as well (vit.py:269-272)
            assert gold["grads"][k] is None
            continue
        assert p.grad is not None and p.grad.dtype == param_dtype, k
        g, r = p.grad.float().cpu().flatten(), leaves[k].grad.flatten()
        rn = float(r.norm())
        assert abs(rn - gold["grads"][k]["l2"]) <= 1e-3 * rn + 1e-9      # oracle == reference fixture
        if rn < 1e-7:
            continue
        cos = float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30))
        assert cos >= 0.99, (k, cos)
        assert abs(float(g.norm()) / rn - 1) <= 5e-2, (k, float(g.norm()), rn)


def test_eval_is_deterministic_and_batch_independent():
    cfg, batch = MODEL_CASES["hilbert224_1d"]
    model = build_model(cfg)
    load_formula(model, cfg)
    model = model.to("cuda", dtype=torch.bfloat16).eval()
    x = formula.image_batch(4, cfg.in_channels, cfg.img_size, cfg.img_size).cuda()
    with torch.no_grad():
        a, b = model(x), model(x)
        assert torch.equal(a, b)
        # every image is an independent unit (the property the data-parallel sharding rests on)
        single = torch.cat([model(x[i:i + 1]) for i in range(4)])
    assert torch.allclose(a.float(), single.float(), atol=1e-2 * float(a.float().abs().max()))


def test_tokenizer_linearity_at_full_size():
    # size-independent property at the BASELINE size: tokens(a*x + b*y) = a*tokens(x) + b*tokens(y) - bias terms
    from sfcvit.tokenizers import HilbertEmbedding1D
    torch.manual_seed(0)
    pe = HilbertEmbedding1D(224, 256, 3, 768).to("cuda", dtype=torch.bfloat16)
    x = torch.randn(8, 3, 224, 224, device="cuda")
    y = torch.randn(8, 3, 224, 224, device="cuda")
    with torch.no_grad():
        bias = pe.proj.bias.float()
        tx, ty, txy = pe(x).float() - bias, pe(y).float() - bias, pe(0.5 * x + 0.25 * y).float() - bias
    err = (txy - (0.5 * tx + 0.25 * ty)).abs().max()
    assert err <= 3e-2 * txy.abs().max()
    # permutation check against plain torch indexing with the registered buffer (bit-exact table)
    idx = pe.hilbert_indices
    ref = x[:, :, idx[:, 0], idx[:, 1]].permute(0, 2, 1).reshape(8, 196, 768)
    ref = ref.bfloat16().float() @ pe.proj.weight.float().t() + bias
    with torch.no_grad():
        assert (pe(x).float() - ref).abs().max() <= 2e-2 * ref.abs().max()


@pytest.mark.parametrize("name", ["hier_morton32", "hier_hilbert32_resample", "hier_morton32_d256", "hier_hilbert32_4lvl"])
def test_hierarchical_tokenizer(name, golden_dir):
    """Hierarchical tokenizers against the oracle and the reference's fixture.  The last two cases are inside the fused
    kernel's envelope (csrc/hier_tokenizer.hip: one kernel for gather + level projections + concat + fusion); the
    first two take the composed path: level kernels, sfcvit_hier_resample_concat (the reference's linear resampling +
    concatenation, multi_hilbert.py:33-38, as one HIP kernel -- torch's interpolate / cat are made to raise here), fusion
    GEMM."""
    from oracle.cases import HIER_CASES, HIER_FUSED_CASES
    from sfcvit.tokenizers import HierarchicalHilbertEmbedding, HierarchicalMortonEmbedding
    img, cin, plist, dim, curve, batch = HIER_CASES[name]
    cls = HierarchicalMortonEmbedding if curve == "z" else HierarchicalHilbertEmbedding
    mod = cls(img, cin, plist, dim)
    sd = vit_oracle.hierarchical_state(img, cin, plist, dim, curve)
    mod.load_state_dict(sd)                                   # same keys as the reference module (fixture "keys")
    x = formula.image_batch(batch, cin, img, img)
    ref = vit_oracle.hierarchical_tokens(x, sd, img, plist, curve)
    mod = mod.to("cuda", dtype=torch.bfloat16)
    assert mod._fusable(x.cuda()) == (name in HIER_FUSED_CASES)

    def _absent(*a, **k):
        raise AssertionError("torch interpolate / cat on the tokenizer's forward path")
    keep = torch.nn.functional.interpolate, torch.cat
    torch.nn.functional.interpolate, torch.cat = _absent, _absent
    try:
        got = mod(x.cuda())
    finally:
        torch.nn.functional.interpolate, torch.cat = keep
    got = got.float().cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max() <= 3e-2 * ref.abs().max()
    with open(os.path.join(golden_dir, "hierarchical.json")) as f:
        gold = json.load(f)[name]
    gv = torch.stack([got.flatten()[i] for i in gold["idx"]])
    assert (gv - torch.tensor(gold["val"])).abs().max() <= 3e-2 * ref.abs().max()


@pytest.mark.parametrize("name,xdt,pdt", [("hier_morton32_d256", torch.float32, torch.bfloat16),
                                          ("hier_hilbert32_4lvl", torch.float32, torch.float32),
                                          ("hier_morton32_d256", torch.bfloat16, torch.bfloat16)])
def test_fused_hierarchical_tokenizer_matches_composed_path_and_oracle_gradients(name, xdt, pdt):
    """Both fused forms -- everything in one kernel, and gather + levels + concatenation in one kernel followed by the
    fusion GEMM (the default) -- against the composed path on the same module: the level outputs are rounded to bf16 at
    the same point, so the outputs differ by fp32 summation order only (<= 2 bf16 ulp of the largest value).  Gradients
    of sum(y * r) w.r.t. every level weight / bias and the fusion weight / bias: fused vs composed (cosine >= 0.999)
    and vs the oracle's autograd (cosine >= 0.99, norm 5 %).  Covers 3 levels x 256 (reference default), 4 levels x 64
    with a ragged last row tile (48 rows), fp32 and bf16 images, fp32 and bf16 parameters."""
    from oracle.cases import HIER_CASES
    from sfcvit.tokenizers import HierarchicalHilbertEmbedding, HierarchicalMortonEmbedding
    img, cin, plist, dim, curve, batch = HIER_CASES[name]
    cls = HierarchicalMortonEmbedding if curve == "z" else HierarchicalHilbertEmbedding
    sd = vit_oracle.hierarchical_state(img, cin, plist, dim, curve)
    x = formula.image_batch(batch, cin, img, img)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    ref = vit_oracle.hierarchical_tokens(x, dict(sd, **leaves), img, plist, curve)
    r = formula.wave("cotangent." + name, tuple(ref.shape))
    (ref * r).sum().backward()

    outs, grads = [], []
    for mode in ("one_kernel", "levels_kernel+gemm", "composed"):
        mod = cls(img, cin, plist, dim)
        mod.load_state_dict(sd)
        mod = mod.to("cuda", dtype=pdt)
        xin = x.to("cuda", dtype=xdt)
        assert mod._fusable(xin)
        y = mod.forward_unfused(xin) if mode == "composed" else mod(xin, one_kernel=(mode == "one_kernel"))
        (y.float() * r.cuda()).sum().backward()
        outs.append(y.detach().float().cpu())
        grads.append({k: p.grad.detach().float().cpu() for k, p in mod.named_parameters()})
    scale = float(ref.detach().abs().max())
    cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
    for o, gr in zip(outs[:2], grads[:2]):
        assert (o - outs[2]).abs().max() <= 2 * 2.0 ** -8 * scale
        assert (o - ref.detach()).abs().max() <= 3e-2 * scale
        assert set(gr) == set(leaves)
        for k in gr:
            g, gc, go = gr[k].flatten(), grads[2][k].flatten(), leaves[k].grad.flatten()
            assert cos(g, gc) >= 0.999, (k, cos(g, gc))
            assert cos(g, go) >= 0.99, (k, cos(g, go))
            assert abs(float(g.norm() / go.norm()) - 1.0) <= 5e-2, k


def test_resampling_hierarchical_tokenizer_gradients_against_the_oracle():
    """hier_hilbert32_resample (256 and 64 tokens: the second level is linearly resampled 64 -> 256): gradients of
    sum(y * r) w.r.t. every level weight / bias and the fusion Linear through sfcvit_hier_resample_concat_bwd against the
    oracle's autograd through F.interpolate (cosine >= 0.99, norm within 5 %)."""
    from oracle.cases import HIER_CASES
    from sfcvit.tokenizers import HierarchicalHilbertEmbedding
    name = "hier_hilbert32_resample"
    img, cin, plist, dim, curve, batch = HIER_CASES[name]
    sd = vit_oracle.hierarchical_state(img, cin, plist, dim, curve)
    x = formula.image_batch(batch, cin, img, img)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    ref = vit_oracle.hierarchical_tokens(x, dict(sd, **leaves), img, plist, curve)
    r = formula.wave("cotangent." + name, tuple(ref.shape))
    (ref * r).sum().backward()
    mod = HierarchicalHilbertEmbedding(img, cin, plist, dim)
    mod.load_state_dict(sd)
    mod = mod.to("cuda", dtype=torch.bfloat16)
    y = mod(x.cuda())
    assert (y.detach().float().cpu() - ref.detach()).abs().max() <= 3e-2 * ref.detach().abs().max()
    (y.float() * r.cuda()).sum().backward()
    for k, p in mod.named_parameters():
        g, go = p.grad.detach().float().cpu().flatten(), leaves[k].grad.flatten()
        cos = float(torch.dot(g, go) / (g.norm() * go.norm() + 1e-30))
        assert cos >= 0.99, (k, cos)
        assert abs(float(g.norm() / go.norm()) - 1.0) <= 5e-2, k


def _tok_cases():
    from oracle.cases import TOKENIZER_CASES
    return sorted(TOKENIZER_CASES)


@pytest.mark.parametrize("name", _tok_cases())
def test_remaining_tokenizers_forward_and_weight_grads(name, golden_dir):
    """Every other tokenizer of the reference's menu through the fused gather+project
    kernel: output vs the oracle and vs the fixture the reference class produced; weight / bias gradients of
    sum(y * r) vs the oracle's autograd."""
    import importlib
    from oracle.cases import TOKENIZER_CASES, RANDPERM_SEED
    modname, clsname, args, kind, batch = TOKENIZER_CASES[name]
    with open(os.path.join(golden_dir, "tokenizers.json")) as f:
        gold = json.load(f)[name]
    mod = getattr(importlib.import_module(modname), clsname)(*args)
    sd = vit_oracle.tokenizer_case_state(args, kind)
    mod.load_state_dict(sd)
    mod = mod.cuda()                                                   # fp32 parameters: exact same values
    x = formula.image_batch(batch, 3, args[0], args[0])

    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    full = dict(sd, **leaves)
    ref = vit_oracle.tokenizer_case_forward(x, full, args, kind, RANDPERM_SEED)
    r = formula.wave("cotangent." + name, tuple(ref.shape))
    (ref * r).sum().backward()

    torch.manual_seed(RANDPERM_SEED)
    got = mod(x.cuda())
    assert list(got.shape) == gold["shape"]
    scale = ref.detach().abs().max()
    assert (got.float().cpu() - ref.detach()).abs().max() <= LOGIT_TOL * scale
    gv = torch.stack([got.float().cpu().flatten()[i] for i in gold["idx"]])
    assert (gv - torch.tensor(gold["val"])).abs().max() <= LOGIT_TOL * scale
    (got.float() * r.cuda()).sum().backward()
    for k, p in mod.named_parameters():
        g, gr = p.grad.float().cpu().flatten(), leaves[k].grad.flatten()
        cos = torch.dot(g, gr) / (g.norm() * gr.norm() + 1e-30)
        assert cos >= 0.99, (k, float(cos))
        assert abs(float(g.norm() / gr.norm()) - 1.0) <= 5e-2, k


def test_reference_default_model_trains_with_head_dim_64():
    """main.py:269-282's model (hierarchical Morton tokenizer, D = 768, depth 8) with 12 heads instead of the
    reference's 4 (head dim 64 is what the attention kernels support): one epoch of the reference-style loop
    on synthetic CIFAR-shaped batches lowers the loss and keeps everything finite."""
    import numpy as np
    from src.models.vit import VisionTransformer1D
    from src.tokenizers.multiscale.multi_morton import HierarchicalMortonEmbedding
    from src.training.train import evaluate, train_with_mixup_or_cutmix
    from sfcvit.training import FusedAdamW, SoftTargetCrossEntropy
    torch.manual_seed(42)
    np.random.seed(42)
    pe = HierarchicalMortonEmbedding(32, 3, [16, 4, 1], 256)
    model = VisionTransformer1D(pe, depth=8, n_heads=12, mlp_dim=512, num_classes=10).to("cuda", dtype=torch.bfloat16)
    opt = FusedAdamW(model.parameters(), lr=3e-4, weight_decay=5e-5)
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(4, 64, 3, 32, 32, generator=g)
    ys = torch.randint(0, 10, (4, 64), generator=g)

    class Loader(list):
        dataset = range(256)
    loader = Loader(zip(xs, ys))
    crit = SoftTargetCrossEntropy()
    losses = [train_with_mixup_or_cutmix(model, loader, crit, opt, None, "cuda")[0] for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    loss, acc = evaluate(model, loader, torch.nn.CrossEntropyLoss(), "cuda")
    assert np.isfinite(loss) and 0.0 <= acc <= 1.0


def test_checkpoint_resume_continues_bit_exactly(tmp_path):
    """main.py:345-354 checkpoint keys; a run resumed from (model_state_dict, optimizer_state_dict) takes the
    same next step as the uninterrupted one (bf16 parameters, fp32 master weights and Adam moments restored)."""
    from sfcvit.training import FusedAdamW, train_step
    cfg, batch = MODEL_CASES["hilbert32_1d"]
    x = formula.image_batch(batch, 3, cfg.img_size, cfg.img_size).cuda()
    tgt = formula.soft_targets(batch, cfg.num_classes).cuda()

    def fresh():
        torch.manual_seed(7)
        m = build_model(cfg).to("cuda", dtype=torch.bfloat16).train()
        return m, FusedAdamW(m.parameters(), lr=1e-3)
    model, opt = fresh()
    for _ in range(2):
        train_step(model, x, tgt, opt)
    path = os.path.join(tmp_path, "ck.pt")
    torch.save({"epoch": 0, "model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict()}, path)
    torch.manual_seed(99)                       # dropout seeds of the next step
    train_step(model, x, tgt, opt)
    want = {k: v.clone() for k, v in model.state_dict().items()}

    model2, opt2 = fresh()
    ck = torch.load(path, map_location="cuda", weights_only=True)
    model2.load_state_dict(ck["model_state_dict"])
    opt2.load_state_dict(ck["optimizer_state_dict"])
    assert opt2.step_count == 2
    torch.manual_seed(99)
    train_step(model2, x, tgt, opt2)
    for k, v in model2.state_dict().items():
        assert torch.equal(v, want[k]), k


def _alt_cases():
    from oracle.cases import ALTVIT_CASES
    return sorted(ALTVIT_CASES)


@pytest.mark.parametrize("name", _alt_cases())
def test_altvit_logits_loss_and_grads(name, golden_dir):
    """SimpleViT / HilbertViT on the HIP kernels vs the fp32 oracle and the reference fixture (same tolerances as the
    main model family)."""
    from oracle.cases import ALTVIT_CASES
    import src.models.altvit as alt
    import sfcvit.functional as F
    clsname, kw, batch = ALTVIT_CASES[name]
    kind = "simple" if clsname == "SimpleViT" else "hilbert"
    with open(os.path.join(golden_dir, "altvit.json")) as f:
        gold = json.load(f)[name]
    sd = vit_oracle.altvit_state(kind, kw["image_size"], kw["patch_size"], kw["dim"], kw["depth"], kw["heads"], kw["mlp_dim"],
                                 kw["num_classes"])
    model = getattr(alt, clsname)(**kw)
    model.load_state_dict(sd)
    model = model.cuda()
    x = formula.image_batch(batch, 3, kw["image_size"], kw["image_size"])
    tgt = formula.soft_targets(batch, kw["num_classes"])
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "pos_embedding"}
    ref_logits = vit_oracle.altvit_forward(x, dict(sd, **leaves), kind, kw["patch_size"], kw["heads"], kw["depth"])
    ref_loss = vit_oracle.soft_target_ce(ref_logits, tgt)
    ref_loss.backward()
    logits = model(x.cuda())
    got = logits.float().cpu()
    scale = ref_logits.detach().abs().max()
    assert (got - ref_logits.detach()).abs().max() <= LOGIT_TOL * scale
    assert (got - torch.tensor(gold["logits"])).abs().max() <= LOGIT_TOL * scale
    loss = F.soft_target_cross_entropy(logits, tgt.cuda())
    assert abs(float(loss.detach()) - gold["loss"]) <= 2e-3 * abs(gold["loss"]) + 2e-3
    loss.backward()
    for k, p in model.named_parameters():
        g, gr = p.grad.float().cpu().flatten(), leaves[k].grad.flatten()
        if float(gr.norm()) < 1e-6:
            continue
        cos = torch.dot(g, gr) / (g.norm() * gr.norm() + 1e-30)
        assert cos >= 0.99, (k, float(cos))
        assert abs(float(g.norm() / gr.norm()) - 1.0) <= 5e-2, k


# ---- BASELINE configurations at their TRUE model dimensions -------------------------------------------
def _full_cases():
    from oracle.cases import FULL_CASES
    return sorted(FULL_CASES)


@pytest.mark.parametrize("name", _full_cases())
def test_full_size_logits_loss_and_grads(name, golden_dir):
    """ViT-Tiny 192/3/12 (batch 256 Hilbert, batch 32 raster), ViT-B 768/12/12 at 224 px with 1000 classes, and the
    ViT-L widths (1024/16 heads/4096, N = 576) at 384 px in z / hilbert / raster order: the HIP path in its production
    precision (bf16 parameters and activations, fp32 accumulation) against the fp32 oracle and the reference fixture.
    Stated tolerances: logits 3e-2 * max|logit| (as the small cases), loss 5e-3 relative, gradient cosine >= 0.99 and
    norm within 5 % for every parameter; tokens 2e-2 * max|token|."""
    import sfcvit.functional as F
    from oracle.cases import FULL_CASES
    from test_oracle_golden import oracle_full_pass
    with open(os.path.join(golden_dir, f"full_{name}.json")) as f:
        gold = json.load(f)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg, batch, sd, leaves, ref_logits, ref_loss = oracle_full_pass(name)
    model = build_model(cfg)
    model.load_state_dict(sd, strict=True)
    if cfg.table_key:
        assert torch.equal(model.state_dict()["patch_embed." + cfg.table_key], sd["patch_embed." + cfg.table_key])
    model = model.to("cuda", dtype=torch.bfloat16).eval()
    x = formula.image_batch(batch, cfg.in_channels, cfg.img_size, cfg.img_size)
    tgt = formula.soft_targets(batch, cfg.num_classes)

    with torch.no_grad():
        tok_ref = vit_oracle.tokenize(x, {k: v.detach() for k, v in sd.items()}, cfg)
        tokens = model.patch_embed(x.cuda())
    assert (tokens.float().cpu() - tok_ref).abs().max() <= 2e-2 * tok_ref.abs().max()

    from sfcvit import ops
    ops.KERNEL_LOG = []
    logits = model(x.cuda())
    got = logits.float().cpu()
    scale = ref_logits.abs().max()
    err = float((got.detach() - ref_logits).abs().max() / scale)
    assert err <= LOGIT_TOL, (name, err)
    gold_logits = torch.tensor(gold["logits"], dtype=torch.float32)
    got_cols = got.detach()[:, gold["logit_cols"]] if "logit_cols" in gold else got.detach()
    assert (got_cols - gold_logits).abs().max() <= LOGIT_TOL * gold_logits.abs().max()
    loss = F.soft_target_cross_entropy(logits, tgt.cuda())
    # loss: 5e-3 relative at these depths (measured 2.5e-3 at ViT-B: 12 bf16 post-LN layers, 1000 classes; log-softmax is
    # 2-Lipschitz in the logits, so the logit tolerance above would allow far more)
    dl = abs(float(loss.detach()) - gold["loss"])
    assert dl <= 5e-3 * abs(gold["loss"]) + 2e-3, (name, dl)
    loss.backward()
    ran, ops.KERNEL_LOG = set(ops.KERNEL_LOG), None
    if name == "vit_b_hilbert224_b64":
        # M = 64 * 196 rows: every encoder GEMM on the persistent 8-phase kernel with 224-row tiles (eval: no dropout bit),
        # weight gradients on its k-major form, attention on the 13-fragment whole-sequence kernels -- what bench.py times
        want = {"gemm8p_kernel<7, 0, true>", "gemm8p_kernel<7, 4, true>", "gemm8p_kernel<7, 33, true>", "gemm8p_kernel<7, 56, true>",
                "gemm8p_km_kernel<true>", "attn_seq_fwd_kernel<13, false>", "attn_seq_bwd_fused_kernel<13, false>"}
        import re
        strip = lambda names: {re.sub(r"gemm8p_kernel<\d, ", "gemm8p_kernel<*, ", k) for k in names}    # noqa: E731  (tile height: dispatcher's choice)
        assert strip(want) <= strip(ran), (sorted(strip(want) - strip(ran)), sorted(ran))
    if name == "vit_l_hilbert384_b16":
        # M = 16 * 576 = 9 216 rows at D = 1 024 / 16 heads / N = 576: the persistent GEMM both ways, the column-sum
        # LayerNorm backward of that width, the long-sequence attention kernels
        _assert_vit_l_kernels(ran)
    worst = (1.0, "")
    for k, p in model.named_parameters():
        if k.startswith("mlp_mixer.token_mix"):
            assert p.grad is None and gold["grads"][k] is None, k
            continue
        g, r = p.grad.float().cpu().flatten(), leaves[k].grad.flatten()
        rn = float(r.norm())
        # (oracle == fixture is pinned at 5e-4 by tests/test_oracle_golden.py on the authoring CPU; the GPU box's host
        # CPU sums in another order)
        assert abs(rn - gold["grads"][k]["l2"]) <= 2e-2 * rn + 1e-9, (k, rn, gold["grads"][k]["l2"])
        if rn < 1e-7:
            continue
        cos = float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30))
        worst = min(worst, (cos, k))
        assert cos >= 0.99, (k, cos)
        assert abs(float(g.norm()) / rn - 1) <= 5e-2, (k, float(g.norm()), rn)
    print(f"[full-size parity] {name}: max|dlogit|/max|logit| = {err:.2e}, |dloss|/loss = {dl / abs(gold['loss']):.2e}, worst gradient cosine = {worst[0]:.5f} ({worst[1]})")

    # ---- the same HIP results against the oracle's rounding-point mode: bf16 at the HIP path's stores, fp32
    # accumulation.  The asked-for 5e-3 * max|logit| / cosine 0.999 does NOT hold end to end, for any implementation whose
    # fp32 sums run in another order: 5e-3 * max|logit| is one bf16 step of the largest logit, and
    # test_every_bf16_store_point_against_the_rounding_point_oracle shows where the rest comes from -- every store
    # replicates to <= 6e-4 of its elements, yet those one-step flips, mixed by the next dense GEMM, put ~10 % of an encoder
    # layer's outputs one step off: a second draw of the rounding noise, as large as the first.  Measured (profiles/r4/
    # rounding_point_parity.txt): logits 4.6e-3 .. 1.6e-2 (fp32 oracle: 6.8e-3 .. 2.1e-2), worst gradient cosine
    # 0.9965 .. 0.9992 (fp32 oracle: 0.9913 .. 0.9979).  Bars: 2e-2 and 0.995, i.e. tighter than the fp32 comparison's
    # 3e-2 / 0.99 by what the shared part of the noise explains; the bit-level bar is the per-store test.
    _, _, _, rleaves, r_logits, r_loss = oracle_full_pass(name, rounded=True)
    r_err = float((got.detach() - r_logits).abs().max() / r_logits.abs().max())
    r_worst = (1.0, "")
    for k, p in model.named_parameters():
        if k.startswith("mlp_mixer.token_mix"):
            continue
        g, r = p.grad.float().cpu().flatten(), rleaves[k].grad.flatten()
        if float(r.norm()) < 1e-7:
            continue
        r_worst = min(r_worst, (float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30)), k))
    print(f"[rounding-point parity] {name}: max|dlogit|/max|logit| = {r_err:.2e}, |dloss| = {abs(float(loss.detach()) - float(r_loss)):.2e}, "
          f"worst gradient cosine = {r_worst[0]:.5f} ({r_worst[1]})")
    assert r_err <= R_LOGIT_TOL, (name, r_err)
    # ... and the HIP path is no further from the fp32 oracle than bf16 storage alone puts the rounding-point oracle (plus a
    # small allowance): the fp32 bars above (3e-2, cosine 0.99) are loose for ViT-L and tight for ViT-Tiny only because the
    # forward's rounding noise differs by model; this one calibrates itself
    o_err = float((r_logits - ref_logits).abs().max() / scale)
    assert err <= o_err + 4e-3, (name, err, o_err)                    # measured: +1.2e-3 at most (profiles/r4/rounding_point_parity.txt)
    margin = (1.0, "")
    for k, p in model.named_parameters():
        if k.startswith("mlp_mixer.token_mix"):
            continue
        r32, rr = leaves[k].grad.flatten(), rleaves[k].grad.flatten()
        if float(r32.norm()) < 1e-7:
            continue
        g = p.grad.float().cpu().flatten()
        cos_o = float(torch.dot(rr, r32) / (rr.norm() * r32.norm() + 1e-30))
        cos_h = float(torch.dot(g, r32) / (g.norm() * r32.norm() + 1e-30))
        margin = min(margin, (cos_h - cos_o, k))
        assert cos_h >= cos_o - 2e-3, (name, k, cos_h, cos_o)          # measured: -1.0e-3 at most
    print(f"[self-calibrated] {name}: logits HIP vs fp32 {err:.2e}, rounding-point oracle vs fp32 {o_err:.2e}; smallest "
          f"(HIP cosine - oracle cosine) against fp32 = {margin[0]:+.4f} ({margin[1]})")
    assert abs(float(loss.detach()) - float(r_loss)) <= 2e-3 * abs(float(r_loss)) + 1e-3, (name, float(loss.detach()), float(r_loss))
    assert r_worst[0] >= R_COS, (name, r_worst)


def _store_point(tag, got, ref, report, max_frac, slack=None):
    """One bf16 store of the HIP path against the rounding-point oracle evaluated on the SAME (HIP-produced) inputs: equal
    bit for bit except where a last-bit difference of the fp32 accumulator crossed a bf16 rounding boundary -- then the two
    are adjacent bf16 values.  Returns nothing; appends (tag, fraction differing) to `report`."""
    got, ref = got.detach().float().cpu().flatten(), ref.detach().float().flatten()
    assert got.shape == ref.shape, (tag, got.shape, ref.shape)
    d = (got - ref).abs()
    frac = float((d > 0).float().mean())
    # adjacent bf16 values are at most 2^-7 of the larger apart; values that cancelled to ~0 carry the accumulator's
    # absolute noise instead (1e-5 of the tensor's largest magnitude covers K = 12 544 terms)
    bound = 2.0 ** -7 * torch.maximum(got.abs(), ref.abs()) + 1e-5 * ref.abs().max()
    if slack is not None:
        bound = bound + slack.flatten()
    worst = float((d / bound).max())
    report.append((tag, frac, worst))
    assert worst <= 1.0, (tag, "more than one bf16 step apart", worst)
    assert frac * d.numel() <= max_frac * d.numel() + 8, (tag, "fraction of elements differing", frac)     # (+8: the logits of 2 images are 2 000 values)


@pytest.mark.parametrize("name", _full_cases())
def test_every_bf16_store_point_against_the_rounding_point_oracle(name):
    """VERDICT r3 #3, in the form that can hold.  The oracle's rounding-point mode rounds to bf16 exactly where the HIP path
    stores bf16.  End to end the two still drift apart (test_full_size_logits_loss_and_grads prints by how much): a
    last-bit difference of an fp32 accumulator flips ~1e-4 of the roundings of a store, the next dense GEMM mixes those
    one-step errors into all of its outputs, and within one encoder layer ~10 % of the elements are one bf16 step off
    (measured at ViT-B: every store below replicates to <= 4e-4 differing, the layer as a whole to 0.128) -- the same
    energy as the rounding noise itself, whatever the implementation.  So the bit-level statement is made per store: every
    intermediate the product's autograd Functions keep (their saved tensors: the real composition, not a re-enactment)
    is compared with the oracle's value computed FROM THE HIP PATH'S OWN INPUTS to that store.  Bar: at most 1e-3 of a
    store's elements differ (2e-3 after a GELU / the K = 12 544 head contraction; measured: <= 6e-4 everywhere, all 94 stores
    of the 12-layer cases), and none by more than one bf16 step.  The attention output is two roundings deep and is
    given the matching allowance (see there); the online-softmax kernels of N = 576 round P against the running maximum,
    so 6-11 % of their outputs land one step away -- bounded by the same allowance, fraction <= 0.25."""
    from oracle.cases import FULL_CASES
    cfg, batch = FULL_CASES[name]
    sd = vit_oracle.formula_state(cfg)
    model = build_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda", dtype=torch.bfloat16).eval()
    x = formula.image_batch(batch, cfg.in_channels, cfg.img_size, cfg.img_size)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    r = lambda t: t.to(torch.bfloat16).float()                                      # noqa: E731
    w = {k: r(v) if torch.is_floating_point(v) else v for k, v in sd.items()}        # the product's parameters are bf16
    cpu = lambda t: t.detach().float().cpu()                                          # noqa: E731
    ln, gelu = vit_oracle.layer_norm, vit_oracle.gelu_erf
    rep = []
    long_seq = cfg.n_patches > 256          # online-softmax kernels: P is rounded against the running maximum
    with vit_oracle.rounding_points():
        tok_ref = vit_oracle.tokenize(x, w, cfg)
    tokens = model.patch_embed(x.cuda())
    _store_point("tokens", tokens, tok_ref, rep, 1e-3)

    t_in = tokens.detach().requires_grad_(True)
    m = model.mlp_mixer(t_in)
    x2, _, _, z, u, h = m.grad_fn.saved_tensors[:6]
    p = "mlp_mixer."
    _store_point("mixer LN", z, r(ln(cpu(x2), w[p + "channel_mix_ln.weight"], w[p + "channel_mix_ln.bias"])), rep, 1e-3)
    pre = cpu(z) @ w[p + "channel_mix.0.weight"].t() + w[p + "channel_mix.0.bias"]
    _store_point("mixer pre-GELU", u, r(pre), rep, 1e-3)
    stored = vit_oracle._pre_gelu_is_stored(z.shape[0], w[p + "channel_mix.0.weight"].shape[0], z.shape[1])
    _store_point("mixer GELU", h, r(gelu(cpu(u) if stored else pre)), rep, 2e-3)
    _store_point("mixer out", m, r(cpu(h) @ w[p + "channel_mix.2.weight"].t() + w[p + "channel_mix.2.bias"] + cpu(x2)).view(m.shape), rep, 1e-3)

    hcur = m.detach()
    B, N, D = hcur.shape
    H, hd = cfg.n_heads, D // cfg.n_heads
    for l in range(cfg.depth):
        p = f"encoder.transformer.layers.{l}."
        layer = model.encoder.transformer.layers[l]
        a = layer.self_attn
        y = F_encoder_layer(hcur.requires_grad_(True), layer, a, cfg.n_heads)
        x2, qkv, o, _, s1, _, _, x1, hh, s2 = y.grad_fn.saved_tensors[:10]
        tag = f"layer {l} "
        _store_point(tag + "qkv", qkv, r(cpu(x2) @ w[p + "self_attn.in_proj_weight"].t() + w[p + "self_attn.in_proj_bias"]), rep, 1e-3)
        qf = cpu(qkv).view(B, N, 3 * D)
        q, k, v = (qf[..., i * D:(i + 1) * D].reshape(B, N, H, hd).transpose(1, 2) for i in range(3))
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        e = torch.exp(s - s.amax(-1, keepdim=True))
        o_ref = r(((r(e) @ v) / e.sum(-1, keepdim=True)).transpose(1, 2).reshape(B * N, D))
        # two roundings deep (P, then the output), and sum_k p v cancels: the output's own step is not the scale of its
        # error.  Allowed on top: every probability of the row one bf16 step off, 2^-8 sum_k p |v| -- the whole-sequence
        # kernels flip ~1e-4 of them (v_exp_f32 vs torch.exp), the online-softmax kernels (N > 256) round P against the
        # running maximum, i.e. draw all of them afresh
        p_abs = ((e / e.sum(-1, keepdim=True)) @ v.abs()).transpose(1, 2).reshape(B * N, D)
        _store_point(tag + "attention", o.reshape(B * N, D), o_ref, rep, 0.25 if long_seq else 2e-3, slack=2.0 ** -8 * p_abs)
        _store_point(tag + "out_proj + x", s1, r(cpu(o).view(B * N, D) @ w[p + "self_attn.out_proj.weight"].t() + w[p + "self_attn.out_proj.bias"] + cpu(x2)), rep, 1e-3)
        _store_point(tag + "LN1", x1, r(ln(cpu(s1), w[p + "norm1.weight"], w[p + "norm1.bias"])), rep, 1e-3)
        _store_point(tag + "relu(linear1)", hh, r(torch.relu(cpu(x1) @ w[p + "linear1.weight"].t() + w[p + "linear1.bias"])), rep, 1e-3)
        _store_point(tag + "linear2 + x1", s2, r(cpu(hh) @ w[p + "linear2.weight"].t() + w[p + "linear2.bias"] + cpu(x1)), rep, 1e-3)
        _store_point(tag + "LN2", y.view(B * N, D), r(ln(cpu(s2), w[p + "norm2.weight"], w[p + "norm2.bias"])), rep, 1e-3)
        hcur = y.detach()

    lg = model.mlp_head(hcur.requires_grad_(True))
    node = lg.grad_fn
    while type(node).__name__ != "_HeadBackward":          # (the class count may be sliced off the padded logits)
        node = node.next_functions[0][0]
    x2, _, _, z, hh, y1, act = node.saved_tensors[:7]
    p = "mlp_head."
    _store_point("head LN", z, r(ln(cpu(x2), w[p + "0.weight"], w[p + "0.bias"])), rep, 1e-3)
    _store_point("head W_emb", hh, r(cpu(z) @ w[p + "1.W_emb"].t()), rep, 1e-3)
    R = w[p + "1.W_emb"].shape[0]
    _store_point("head W_seq", y1, r(cpu(hh).view(B, N * R) @ w[p + "1.W_seq"].reshape(-1, N * R).t()), rep, 2e-3)
    _store_point("head GELU", act, r(gelu(cpu(y1))), rep, 2e-3)
    _store_point("logits", lg, r(cpu(act) @ w[p + "4.weight"].t() + w[p + "4.bias"]), rep, 1e-3)
    top = sorted(rep, key=lambda t: -t[1])[:3]
    print(f"[store points] {name}: {len(rep)} stores, differing fraction max {top[0][1]:.2e} ({top[0][0]}), then "
          f"{top[1][1]:.2e} ({top[1][0]}), {top[2][1]:.2e} ({top[2][0]}); worst distance {max(t[2] for t in rep):.2f} bf16 steps")


def F_encoder_layer(h, layer, a, n_heads):
    import sfcvit.functional as F
    return F.encoder_layer(h, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, layer.norm1.weight,
                           layer.norm1.bias, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
                           layer.norm2.weight, layer.norm2.bias, n_heads, layer.norm1.eps)


def _assert_vit_l_kernels(ran):
    import re
    strip = lambda names: {re.sub(r"gemm8p_kernel<\d, ", "gemm8p_kernel<*, ", k) for k in names}    # noqa: E731  (tile height: dispatcher's choice)
    want = {"gemm8p_kernel<7, 0, true>", "gemm8p_kernel<7, 4, true>", "gemm8p_kernel<7, 33, true>", "gemm8p_kernel<7, 56, true>",
            "gemm8p_km_kernel<true>", "ln_bwd_cols_kernel<4, true>", "attn_long_fwd_kernel<36>", "attn_long_bwd_kv_kernel"}
    assert strip(want) <= strip(ran), (sorted(strip(want) - strip(ran)), sorted(ran))
    return sorted({k[14] for k in ran if k.startswith("gemm8p_kernel<")})


def test_vit_l_batch64_equals_its_four_batch16_shards():
    """The twin of the ViT-B test below at configuration 5's geometry (ViT-L/16 at 384 px: D = 1 024, 16 heads, N = 576
    tokens, mlp 4 096; 4 of the 24 layers to keep the test short -- every layer launches the same kernels): batch 64
    (M = 36 864 rows) against the same images as four batches of 16 (M = 9 216).  Eval mode; the sides differ in launch
    geometry only.  The kernels asserted are the ones a ViT-L step runs on: the persistent GEMM (forward and dX, with bias /
    ReLU bit-mask / column-sum epilogues) and its k-major weight-gradient form, ln_bwd_cols_kernel at D = 1 024, the
    long-sequence attention kernels at 36 key fragments."""
    import sfcvit.functional as F
    from sfcvit import ops
    from oracle.vit_oracle import OracleConfig
    cfg = OracleConfig("hilbert1d", 384, 256, 3, 1024, depth=4, n_heads=16, mlp_dim=4096, num_classes=1000, variant="1d")
    model = build_model(cfg)
    load_formula(model, cfg)
    model = model.to("cuda", dtype=torch.bfloat16).eval()
    x = formula.image_batch(64, cfg.in_channels, cfg.img_size, cfg.img_size).cuda()
    tgt = formula.soft_targets(64, cfg.num_classes).cuda()
    params = [(k, p) for k, p in model.named_parameters() if not k.startswith("mlp_mixer.token_mix")]

    def run(xb, tb):
        for _, p in params:
            p.grad = None
        ops.KERNEL_LOG = []
        logits = model(xb)
        loss = F.soft_target_cross_entropy(logits, tb)
        loss.backward()
        ran, ops.KERNEL_LOG = set(ops.KERNEL_LOG), None
        return logits.detach().float(), float(loss.detach()), [p.grad.detach().float().clone() for _, p in params], ran

    full_logits, full_loss, full_grads, ran = run(x, tgt)
    heights = _assert_vit_l_kernels(ran)
    shard_logits, shard_losses, shard_grads = [], [], None
    for i in range(4):
        lg, ls, gr, ran16 = run(x[16 * i:16 * i + 16], tgt[16 * i:16 * i + 16])
        shard_logits.append(lg)
        shard_losses.append(ls)
        shard_grads = gr if shard_grads is None else [a + b for a, b in zip(shard_grads, gr)]
    heights16 = _assert_vit_l_kernels(ran16)
    cat = torch.cat(shard_logits)
    assert torch.isfinite(full_logits).all()
    assert (full_logits - cat).abs().max() <= 1e-2 * cat.abs().max()        # same rows through the same kernels
    assert abs(full_loss - sum(shard_losses) / 4) <= 1e-3 * abs(full_loss)
    worst = (1.0, "")
    for (k, _), gf, gs in zip(params, full_grads, shard_grads):
        a, b = gf.flatten(), gs.flatten() / 4
        if float(b.norm()) < 1e-9:
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        worst = min(worst, (cos, k))
        assert cos >= 0.995, (k, cos)
        assert abs(float(a.norm() / b.norm()) - 1) <= 2e-2, (k, float(a.norm()), float(b.norm()))
    print(f"[vit-l shards] gemm8p tile heights (x32 rows) at M = 36 864: {heights}, at M = 9 216: {heights16}; "
          f"worst gradient cosine batch 64 vs 4 x 16 = {worst[0]:.5f} ({worst[1]})")


def test_vit_b_batch256_equals_its_four_batch64_shards():
    """The data-parallel property at the benched M (VERDICT r2 #1c): ViT-B/16@224 Hilbert at batch 256 (M = 50 176 rows,
    the launch geometry bench.py times) against the same images as four batches of 64: per-image logits, the loss (=
    mean of the shard losses) and every parameter gradient (= mean of the shard gradients).  Eval mode, so nothing is
    random; the two sides differ only in launch geometry (tile rounds, split-K of the head and weight-gradient GEMMs),
    i.e. in fp32 summation order and bf16 rounding of the gradient sums."""
    import sfcvit.functional as F
    from oracle.cases import FULL_CASES
    cfg, _ = FULL_CASES["vit_b_hilbert224"]
    model = build_model(cfg)
    load_formula(model, cfg)
    model = model.to("cuda", dtype=torch.bfloat16).eval()
    x = formula.image_batch(256, cfg.in_channels, cfg.img_size, cfg.img_size).cuda()
    tgt = formula.soft_targets(256, cfg.num_classes).cuda()
    params = [(k, p) for k, p in model.named_parameters() if not k.startswith("mlp_mixer.token_mix")]

    def run(xb, tb):
        for _, p in params:
            p.grad = None
        logits = model(xb)
        loss = F.soft_target_cross_entropy(logits, tb)
        loss.backward()
        return logits.detach().float(), float(loss.detach()), [p.grad.detach().float().clone() for _, p in params]

    full_logits, full_loss, full_grads = run(x, tgt)
    shard_logits, shard_losses, shard_grads = [], [], None
    for i in range(4):
        lg, ls, gr = run(x[64 * i:64 * i + 64], tgt[64 * i:64 * i + 64])
        shard_logits.append(lg)
        shard_losses.append(ls)
        shard_grads = gr if shard_grads is None else [a + b for a, b in zip(shard_grads, gr)]
    cat = torch.cat(shard_logits)
    assert (full_logits - cat).abs().max() <= 1e-2 * cat.abs().max()        # same rows through the same kernels
    assert abs(full_loss - sum(shard_losses) / 4) <= 1e-3 * abs(full_loss)
    for (k, _), gf, gs in zip(params, full_grads, shard_grads):
        a, b = gf.flatten(), gs.flatten() / 4
        if float(b.norm()) < 1e-9:
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        assert cos >= 0.995, (k, cos)
        assert abs(float(a.norm() / b.norm()) - 1) <= 2e-2, (k, float(a.norm()), float(b.norm()))


def _hip_train_run(name, zero_to_none=True, with_reducer=False):
    """sfcvit.training.train_step from the formula state of a TRAIN_CASES entry -> (losses, grad norms, named fp32
    master weights, model)."""
    from oracle.cases import TRAIN_CASES
    from sfcvit.training import FusedAdamW, GradReducer, train_step
    case, steps, lr, wd = TRAIN_CASES[name]
    cfg, batch = MODEL_CASES[case]
    model = build_model(cfg)
    load_formula(model, cfg)
    model = model.to("cuda", dtype=torch.bfloat16).eval()           # eval(): dropout off, gradients flow
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=wd, max_grad_norm=1.0)
    if not zero_to_none:
        zg = opt.zero_grad
        opt.zero_grad = lambda: zg(set_to_none=False)
    reducer = GradReducer(opt) if with_reducer else None
    x = formula.image_batch(batch, cfg.in_channels, cfg.img_size, cfg.img_size).cuda()
    tgt = formula.soft_targets(batch, cfg.num_classes).cuda()
    losses, norms = [], []
    for _ in range(steps):
        losses.append(float(train_step(model, x, tgt, opt, reducer=reducer)))
        norms.append(float(opt.grad_norm()))
    names = {id(p): k for k, p in model.named_parameters()}
    master = {names[id(p)]: opt.master[o:o + p.numel()].view(p.shape).cpu() for p, o in zip(opt.active, opt.offsets)}
    return losses, norms, master, model


@pytest.mark.parametrize("name", ["hilbert32_1d", "raster32_2d"])
def test_train_steps_match_oracle_and_reference(name, golden_dir):
    """Three whole optimisation steps (src/training/train.py:153-167: zero_grad -> forward -> soft-target CE -> backward
    -> clip 1.0 -> AdamW) of sfcvit.training.train_step -- gradients written in place into the flat bf16 buffer, fused
    clip + fp32-master AdamW -- against oracle.vit_oracle.train_step and the fixture from the reference model with
    torch.optim.AdamW.  Tolerances (bf16 forward/backward vs fp32): per-step loss 1e-2 relative (+5e-3 per further step); pre-clip gradient
    norm 3 % (+1 % per further step); the UPDATE (master weight after 3 steps minus initial value) of every parameter with a non-negligible
    gradient: cosine >= 0.9 with the oracle's update and every element within 2 * lr * steps (Adam's maximum drift).

    TRAIN_CASES also holds a non-overshooting rate (hilbert32_1d_lr1e4): see
    test_train_steps_at_a_small_rate_follow_the_bf16_working_weight_oracle."""
    from oracle.cases import TRAIN_CASES
    from test_oracle_golden import oracle_train_run
    with open(os.path.join(golden_dir, f"train_{name}.json")) as f:
        gold = json.load(f)
    case, steps, lr, wd = TRAIN_CASES[name]
    cfg, _ = MODEL_CASES[case]
    ref_losses, _, ref_sd = oracle_train_run(name)
    init = vit_oracle.formula_state(cfg)
    losses, norms, master, model = _hip_train_run(name)
    for s in range(steps):
        # the trajectories separate step by step (lr 1e-3 on a 4-image batch overshoots: 2.54 -> 1.06 -> 1.24 in the
        # reference itself): 1 % at the first step, +0.5 % per further step
        tol = 1e-2 + 5e-3 * s
        assert abs(losses[s] - ref_losses[s]) <= tol * abs(ref_losses[s]) + 2e-3, (s, losses, ref_losses)
        assert abs(losses[s] - gold["loss"][s]) <= tol * abs(gold["loss"][s]) + 2e-3
        # the pre-clip norm swings 13.9 -> 4.7 -> 9.2 along that overshooting trajectory, so it is as sensitive to the
        # position reached as the loss is: 3 % at the first step, +1 % per further step (observed 0.3 / 1.0 / 3.5 %)
        assert abs(norms[s] / gold["grad_norm"][s] - 1) <= 3e-2 + 1e-2 * s, (s, norms, gold["grad_norm"])
    assert set(master) == {k for k in gold["params"] if not k.startswith("mlp_mixer.token_mix")}
    # Adam normalises every element's step to ~lr whatever its gradient's size, so elements whose gradient is rounding
    # noise (the key third of in_proj_bias is exactly zero in exact arithmetic) move by +-lr at random in ANY
    # implementation.  Compared are therefore the CONFIDENT elements: those the oracle moved by >= 0.8 * lr * steps
    # (same gradient sign in every step).  A wrong slot offset / stale gradient gives ~50 % sign agreement there.
    confident = total = 0
    for k, w in master.items():
        # the HIP run starts from the bf16-rounded parameters (model.to(bfloat16), as main.py:157 does): its update is
        # measured from there, the oracle's from the fp32 value
        upd, ref_upd = (w - init[k].bfloat16().float()).flatten(), (ref_sd[k].detach() - init[k]).flatten()
        assert (upd - ref_upd).abs().max() <= 2.1 * lr * steps, k
        mask = ref_upd.abs() >= 0.8 * lr * steps
        confident += int(mask.sum())
        total += mask.numel()
        if int(mask.sum()) >= 8:
            agree = float((torch.sign(upd[mask]) == torch.sign(ref_upd[mask])).float().mean())
            assert agree >= 0.9, (k, agree)
            assert float((upd[mask] - ref_upd[mask]).abs().mean()) <= 0.25 * lr * steps, k
        # the bf16 working copy is the rounded master
        assert torch.equal(dict(model.named_parameters())[k].detach().cpu(), w.to(torch.bfloat16)), k
    assert confident >= 0.1 * total, (confident, total)          # the comparison is not vacuous (measured: 26 %)


def test_train_steps_at_a_small_rate_follow_the_bf16_working_weight_oracle():
    """Four steps at lr 1e-4 (TRAIN_CASES hilbert32_1d_lr1e4: no overshoot) against the oracle run with the HIP path's weight
    handling -- fp32 master, bf16 working copy (test_oracle_golden.oracle_train_run_bf16_working).  With the lag of the working
    copy modelled, the trajectories stay together and the tolerances need no widening: loss 1 % and pre-clip gradient norm
    2.5 % at EVERY step (the all-fp32 reference is 13 % away at step 2)."""
    from test_oracle_golden import oracle_train_run_bf16_working
    name = "hilbert32_1d_lr1e4"
    ref_losses, ref_norms = oracle_train_run_bf16_working(name)
    losses, norms, _, _ = _hip_train_run(name)
    for s, (a, b) in enumerate(zip(losses, ref_losses)):
        assert abs(a - b) <= 1e-2 * b, (s, losses, ref_losses)
    for s, (a, b) in enumerate(zip(norms, ref_norms)):
        assert abs(a / b - 1) <= 2.5e-2, (s, norms, ref_norms)


def test_train_step_variants_are_bit_identical():
    """The same three steps with zero_grad(set_to_none=False) (gradients accumulate into existing views instead of
    being adopted in place) and with a world-size-1 GradReducer (hooks + bucket bookkeeping, no collective) must
    reproduce the plain run bit for bit: same kernels, same order, different plumbing."""
    base = _hip_train_run("hilbert32_1d")
    keep = _hip_train_run("hilbert32_1d", zero_to_none=False)
    red = _hip_train_run("hilbert32_1d", with_reducer=True)
    for other in (keep, red):
        assert other[0] == base[0] and other[1] == base[1]
        for k in base[2]:
            assert torch.equal(other[2][k], base[2][k]), k


def test_weight_transposes_of_a_backward_pass_come_from_one_launch():
    """The dX GEMMs read W^T.  Inside a backward pass the first request transposes every 2-D weight of the flat buffer
    in ONE launch (FlatGradBuffer.transposed, keyed by autograd's graph-task id); per-GEMM transposes are the fallback
    (SFCVIT_WT_CACHE=0, parameters outside a flat buffer, the first step).  Both must train bit-identically."""
    import sfcvit.training.optim as optim
    from sfcvit import ops
    from sfcvit.models import VisionTransformer1D
    from sfcvit.tokenizers import HilbertEmbedding1D
    from sfcvit.training import FusedAdamW, mixup_soft_targets, train_step

    def run(cache):
        optim._WT_CACHE = cache
        calls = {"single": 0, "batched": 0}
        single, batched = ops.transpose, ops.transpose_batched
        ops.transpose = lambda x: (calls.__setitem__("single", calls["single"] + 1), single(x))[1]
        ops.transpose_batched = lambda *a: (calls.__setitem__("batched", calls["batched"] + 1), batched(*a))[1]
        try:
            torch.manual_seed(11)
            model = VisionTransformer1D(HilbertEmbedding1D(32, 16, 3, 256), depth=2, n_heads=4, mlp_dim=512, num_classes=16,
                                        dropout_p=0.1, head_dropout_p=0.5).to("cuda", dtype=torch.bfloat16).train()
            opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=5e-5, max_grad_norm=1.0)
            g = torch.Generator(device="cuda").manual_seed(5)
            x = torch.randn(8, 3, 32, 32, device="cuda", generator=g)
            t = mixup_soft_targets(torch.randint(0, 16, (8,), device="cuda", generator=g), 16, lam=0.7)
            losses = [float(train_step(model, x, t, opt)) for _ in range(3)]
            return losses, opt.master.clone(), calls
        finally:
            ops.transpose, ops.transpose_batched = single, batched
            optim._WT_CACHE = True
    l1, m1, c1 = run(True)
    l0, m0, c0 = run(False)
    assert c0["batched"] == 0 and c0["single"] > 0
    # step 1 builds the flat buffers in optimizer.step(), i.e. after its backward: steps 2 and 3 use the batch
    assert c1["batched"] == 2 and c1["single"] == c0["single"] // 3, (c1, c0)
    assert l1 == l0 and torch.equal(m1, m0)


def test_two_stage_and_fused_patch_embed_agree():
    """Default: sfcvit_tokens_gather + the projection on the GEMM kernels (functional._PatchEmbed2); SFCVIT_PE_FUSED=1: the fused
    gather-GEMM kernels of rounds 1-2.  Same tokens (one bf16 rounding of the same pixels), same products, another summation
    order: embeddings within bf16 rounding, weight gradients within the split-K tolerance."""
    import sfcvit.functional as F
    from sfcvit.tokenizers import HilbertEmbedding1D
    torch.manual_seed(4)
    tok = HilbertEmbedding1D(224, 256, 3, 768).to("cuda", dtype=torch.bfloat16)
    x = torch.randn(6, 3, 224, 224, device="cuda")
    gy = torch.randn(6, 196, 768, device="cuda").to(torch.bfloat16)
    res = {}
    for two in (True, False):
        F.PE_TWO_STAGE = two
        try:
            tok.zero_grad()
            y = tok(x)
            y.backward(gy)
            res[two] = (y.detach().float(), tok.proj.weight.grad.float().clone(), tok.proj.bias.grad.float().clone())
        finally:
            F.PE_TWO_STAGE = True
    for a, b, tol in zip(res[True], res[False], (1 / 128, 1 / 64, 1 / 64)):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())
    assert float(torch.nn.functional.cosine_similarity(res[True][1].flatten(), res[False][1].flatten(), dim=0)) > 0.9999


def test_deferred_reductions_train_bit_identically_and_leave_nothing_queued():
    """Bias / LayerNorm-parameter gradients written into gradient slots have their final reductions queued and launched once at
    the end of the backward pass (ops._Deferring, sfcvit_reduce_flush).  Same kernels, same summation order: the run must
    equal the one with SFCVIT_DEFER_REDUCE=0 bit for bit, gradients must be complete when backward() returns, and the
    queue must be empty afterwards."""
    from sfcvit import ops
    from sfcvit._lib import lib
    from sfcvit.models import VisionTransformer1D
    from sfcvit.tokenizers import HilbertEmbedding1D
    from sfcvit.training import FusedAdamW, mixup_soft_targets, train_step

    def run(defer):
        ops.DEFER_REDUCES = defer
        flushes = {"n": 0, "items": 0}
        flush = lib.sfcvit_reduce_flush

        def counting(stream):
            flushes["n"] += 1
            flushes["items"] += lib.sfcvit_reduce_pending()
            return flush(stream)
        lib.sfcvit_reduce_flush = counting
        try:
            torch.manual_seed(11)
            model = VisionTransformer1D(HilbertEmbedding1D(32, 16, 3, 256), depth=2, n_heads=4, mlp_dim=512, num_classes=16,
                                        dropout_p=0.1, head_dropout_p=0.5).to("cuda", dtype=torch.bfloat16).train()
            opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=5e-5, max_grad_norm=1.0)
            g = torch.Generator(device="cuda").manual_seed(5)
            x = torch.randn(8, 3, 32, 32, device="cuda", generator=g)
            t = mixup_soft_targets(torch.randint(0, 16, (8,), device="cuda", generator=g), 16, lam=0.7)
            losses = [float(train_step(model, x, t, opt)) for _ in range(3)]
            assert lib.sfcvit_reduce_pending() == 0
            # a bare backward: gradients are final when it returns (the engine's end-of-pass callback flushed the queue)
            opt.zero_grad()
            import sfcvit.functional as F
            F.soft_target_cross_entropy(model(x), t).backward()
            assert lib.sfcvit_reduce_pending() == 0
            opt.adopt_all()
            return losses, opt.master.clone(), opt.flat_grad.clone(), flushes
        finally:
            lib.sfcvit_reduce_flush = flush
            ops.DEFER_REDUCES = True
    l1, m1, g1, f1 = run(True)
    l0, m0, g0, f0 = run(False)
    assert f0["n"] == 0
    # the first step has no slots yet (the flat buffers are built by its optimizer step): three later backward passes defer
    assert f1["n"] == 3 and f1["items"] >= 3 * 10, f1
    assert l1 == l0 and torch.equal(m1, m0) and torch.equal(g1, g0)


def test_shared_parameter_gradient_accumulates_once_per_use():
    """A weight used twice in one backward: the first use may claim the in-place gradient slot, the second must be
    ADDED by autograd (functional._slot hands the slot out once per zero_grad epoch)."""
    import sfcvit.functional as F
    from sfcvit.training import FusedAdamW
    torch.manual_seed(3)
    w = torch.nn.Parameter((torch.randn(256, 256) / 16).to("cuda", torch.bfloat16))
    b = torch.nn.Parameter(torch.zeros(256, device="cuda", dtype=torch.bfloat16))
    x = torch.randn(512, 256, device="cuda").to(torch.bfloat16)
    opt = FusedAdamW([w, b], lr=0.0, weight_decay=0.0, max_grad_norm=None)

    def loss_of(wt, bt, xt, lin):
        return lin(lin(xt, wt, bt), wt, bt).float().pow(2).mean()
    loss_of(w, b, x, F.linear).backward()
    opt.step()                                   # builds the flat buffers (lr = 0: values unchanged)
    for none in (True, False):
        opt.zero_grad(set_to_none=none)
        loss_of(w, b, x, F.linear).backward()
        opt.adopt_all()
        wr, br = w.detach().float().requires_grad_(True), b.detach().float().requires_grad_(True)
        loss_of(wr, br, x.float(), torch.nn.functional.linear).backward()
        for got, ref in ((w.grad, wr.grad), (b.grad, br.grad)):
            g, r = got.float().flatten(), ref.flatten()
            assert float(torch.dot(g, r) / (g.norm() * r.norm())) >= 0.999
            assert abs(float(g.norm() / r.norm()) - 1) <= 2e-2
        assert w.grad.data_ptr() == opt.flat_grad.data_ptr() + 2 * opt.offsets[0]


def test_shared_layernorm_and_mixer_parameters_with_deferred_reductions():
    """a LayerNorm gamma / beta (and a mixer's vectors) used TWICE in one backward while their reductions are
    deferred to the end of the pass: the second use makes autograd add (slot view + fresh tensor), which reads the slot --
    functional._slot flushes the queue first.  Gradients must equal the ones computed with deferral switched off (bit for
    bit: same kernels, same order) and agree with torch's fp32 autograd."""
    import sfcvit.functional as F
    from sfcvit import ops
    from sfcvit.training import FusedAdamW
    torch.manual_seed(5)
    D = 256
    dev = "cuda"
    g = torch.nn.Parameter((1 + 0.1 * torch.randn(D)).to(dev, torch.bfloat16))
    be = torch.nn.Parameter((0.1 * torch.randn(D)).to(dev, torch.bfloat16))
    w1 = torch.nn.Parameter((torch.randn(2 * D, D) / 16).to(dev, torch.bfloat16))
    b1 = torch.nn.Parameter((0.1 * torch.randn(2 * D)).to(dev, torch.bfloat16))
    w2 = torch.nn.Parameter((torch.randn(D, 2 * D) / 22).to(dev, torch.bfloat16))
    b2 = torch.nn.Parameter((0.1 * torch.randn(D)).to(dev, torch.bfloat16))
    params = [g, be, w1, b1, w2, b2]
    x = torch.randn(4, 128, D, device=dev).to(torch.bfloat16)
    opt = FusedAdamW(params, lr=0.0, weight_decay=0.0, max_grad_norm=None)

    def loss_hip(xt):
        h = F.layer_norm(xt, g, be)
        h = F.mixer_block(h, g, be, w1, b1, w2, b2)          # gamma / beta tied between the two LayerNorms
        h = F.mixer_block(h, g, be, w1, b1, w2, b2)          # ... and the whole mixer used twice
        return F.layer_norm(h, g, be).float().pow(2).mean()

    def loss_ref(xt, ps):
        gg, bb, a1, c1, a2, c2 = ps
        ln = lambda t: torch.nn.functional.layer_norm(t, (D,), gg, bb, 1e-5)
        mix = lambda t: t + torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(ln(t), a1, c1)), a2, c2)
        return ln(mix(mix(ln(xt)))).pow(2).mean()
    loss_hip(x).backward()
    opt.step()                                   # builds the flat buffers (lr = 0: values unchanged)
    grads = {}
    saved = ops.DEFER_REDUCES
    try:
        for defer in (True, False):
            ops.DEFER_REDUCES = defer
            opt.zero_grad(set_to_none=True)
            loss_hip(x).backward()
            opt.adopt_all()
            torch.cuda.synchronize()
            assert ops.lib.sfcvit_reduce_pending() == 0
            grads[defer] = [p.grad.detach().float().clone() for p in params]
    finally:
        ops.DEFER_REDUCES = saved
    ref = [p.detach().float().requires_grad_(True) for p in params]
    loss_ref(x.float(), ref).backward()
    for name, a, b, r in zip("gamma beta w1 b1 w2 b2".split(), grads[True], grads[False], ref):
        assert torch.equal(a, b), f"{name}: deferred and immediate reductions disagree"
        ga, rr = a.flatten(), r.grad.flatten()
        cos = float(torch.dot(ga, rr) / (ga.norm() * rr.norm()))
        assert cos >= 0.995 and abs(float(ga.norm() / rr.norm()) - 1) <= 3e-2, (name, cos)


def test_graphed_train_step_replays_bit_exactly_and_draws_new_masks():
    """GraphedTrainStep: the whole training step (training mode: dropout 0.1 / 0.5 on) captured into one hipGraph with
    the step state on the device (sfcvit_step_advance) must take the SAME steps as the same code run eagerly in
    device-state mode -- bit for bit, over warm-up + replays -- and two replays must not reuse a dropout mask."""
    from sfcvit import ops
    from sfcvit.training import FusedAdamW, GraphedTrainStep, train_step
    cfg, batch = MODEL_CASES["hilbert32_1d"]
    x = formula.image_batch(batch, 3, cfg.img_size, cfg.img_size).cuda()
    tgt = formula.soft_targets(batch, cfg.num_classes).cuda()

    def fresh():
        m = build_model(cfg)
        load_formula(m, cfg)
        m = m.to("cuda", dtype=torch.bfloat16).train()
        return m, FusedAdamW(m.parameters(), lr=1e-3, weight_decay=5e-2)
    try:
        model_e, opt_e = fresh()
        opt_e.use_device_state(seed_base=4242)
        eager = []
        for _ in range(6):
            eager.append(float(train_step(model_e, x, tgt, opt_e)))       # begin_step() advances the device state
        model_g, opt_g = fresh()
        opt_g.use_device_state(seed_base=4242)
        step = GraphedTrainStep(model_g, x.clone(), tgt.clone(), opt_g, warmup=2, preserve_state=False)   # 2 eager steps that count, then capture
        graphed = [float(step()) for _ in range(4)]
        assert graphed == eager[2:], (graphed, eager)
        assert opt_g.step_count == 6 and int(opt_g.dev_state[1]) == 6
        assert opt_e.step_count == 6 and int(opt_e.dev_state[1]) == 6     # eager steps in device-state mode count on both sides
        for (k, a), (_, b) in zip(model_e.state_dict().items(), model_g.state_dict().items()):
            assert torch.equal(a, b), k
        assert torch.equal(opt_e.master, opt_g.master) and torch.equal(opt_e.v, opt_g.v)
        # same weights, same batch, two consecutive seed offsets: different masks -> different training-mode logits
        with torch.no_grad():
            opt_g.advance(); a = model_g(x).float()
            opt_g.advance(); b = model_g(x).float()
        assert not torch.equal(a, b)
        step.close()                                   # the graphed step owns the device state's lifetime
        assert ops.STEP_STATE is None and opt_g.dev_state is None
        with pytest.raises(RuntimeError, match="closed"):
            step()
    finally:
        ops.STEP_STATE = None


def test_graphed_epoch_loop_trains_like_the_eager_loop_and_leaves_state_untouched_by_its_warm_up():
    """train_with_mixup_or_cutmix(graphed=GraphedTrainStep(...)), main.py --graph: (a) building the graphed step (three
    warm-up steps on the static buffers + capture) leaves parameters, Adam moments, step count and scheduler position
    exactly as they were; (b) an epoch through the graph gives the same loss / accuracy and the same final weights, bit
    for bit, as the eager loop in device-state mode from the same state with the same host RNG (dropout on)."""
    import numpy as np
    from sfcvit import ops
    from sfcvit.training import FusedAdamW, GraphedTrainStep, SoftTargetCrossEntropy, WarmupCosine
    from sfcvit.training.loops import train_with_mixup_or_cutmix
    cfg, _ = MODEL_CASES["hilbert32_1d"]
    g = torch.Generator().manual_seed(0)
    xs, ys = torch.randn(3, 8, 3, 32, 32, generator=g), torch.randint(0, cfg.num_classes, (3, 8), generator=g)
    # ... and a last, smaller batch (a real DataLoader's tail): it cannot replay the graph and runs eagerly IN device-state
    # mode -- its step must advance the device counters and the host mirror like any other
    x_tail, y_tail = torch.randn(5, 3, 32, 32, generator=g), torch.randint(0, cfg.num_classes, (5,), generator=g)
    batches = list(zip(xs, ys)) + [(x_tail, y_tail)]

    class Loader(list):
        dataset = range(29)

    def epoch(graph):
        m = build_model(cfg)
        load_formula(m, cfg)
        m = m.to("cuda", dtype=torch.bfloat16).train()
        o = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=5e-5)
        sch = WarmupCosine(o, 2, 20)
        o.use_device_state(torch.device("cuda"), seed_base=99)
        before = [p.detach().clone() for p in m.parameters()]
        gs = None
        if graph:
            gs = GraphedTrainStep(m, torch.zeros(8, 3, 32, 32, device="cuda"), torch.zeros(8, cfg.num_classes, device="cuda"), o, sch)
            assert all(torch.equal(a, b) for a, b in zip(before, m.parameters()))
            assert o.step_count == 0 and int(o.dev_state[1]) == 0 and sch.n == 0 and not o.m.any() and not o.v.any()
            assert torch.equal(o.master, o.flat_param.float()) and o.lr == sch.lr_at(0)
        torch.manual_seed(11)
        np.random.seed(11)
        out = train_with_mixup_or_cutmix(m, Loader(batches), SoftTargetCrossEntropy(), o, sch, "cuda", graphed=gs)
        assert o.step_count == 4 and int(o.dev_state[1]) == 4 and sch.n == 4      # three replays (or eager steps) + the tail
        return out, [p.detach().float().clone() for p in m.parameters()], o.master.clone()

    try:
        (l_e, a_e), p_e, w_e = epoch(False)
        (l_g, a_g), p_g, w_g = epoch(True)
    finally:
        ops.STEP_STATE = None
    assert l_g == l_e and a_g == a_e
    assert all(torch.equal(a, b) for a, b in zip(p_e, p_g)) and torch.equal(w_e, w_g)


def test_fused_adamw_loads_a_torch_adamw_state_dict():
    """A reference checkpoint's `optimizer_state_dict` is torch.optim.AdamW's (main.py:288-289, 345-354): per-parameter
    step / exp_avg / exp_avg_sq must land in FusedAdamW's flat m / v buffers and step count (parameters without state --
    the unused token-mix branch -- stay outside), and the next step from that state must be the step torch would take."""
    from sfcvit.training import FusedAdamW
    cfg, _ = MODEL_CASES["hilbert32_1d"]
    torch.manual_seed(0)
    model = build_model(cfg).to("cuda", dtype=torch.bfloat16)
    named = [(k, p) for k, p in model.named_parameters()]
    used = [(k, p) for k, p in named if not k.startswith("mlp_mixer.token_mix")]
    ref = torch.optim.AdamW(model.parameters(), lr=3e-4, weight_decay=5e-5)
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(2):                                               # two reference steps on fabricated gradients
        for _, p in used:
            p.grad = (torch.randn(p.shape, device="cuda", generator=g) * 1e-2).to(torch.bfloat16)
        ref.step()
    sd = ref.state_dict()
    before = {k: p.detach().clone() for k, p in named}
    opt = FusedAdamW(model.parameters(), lr=1.0, weight_decay=5e-5, max_grad_norm=None)
    opt.load_state_dict(sd)
    assert opt.step_count == 2 and opt.lr == pytest.approx(3e-4) and len(opt.active) == len(used)
    index = {id(p): i for i, p in enumerate(model.parameters())}
    for p, o in zip(opt.active, opt.offsets):
        st = sd["state"][index[id(p)]]
        assert torch.equal(opt.m[o:o + p.numel()], st["exp_avg"].flatten().float())
        assert torch.equal(opt.v[o:o + p.numel()], st["exp_avg_sq"].flatten().float())
    assert all(torch.equal(p, before[k]) for k, p in named)          # weights untouched by loading the state
    # third step, same gradients on both sides: torch updates the bf16 parameters in place, FusedAdamW its fp32 master
    grads = [(torch.randn(p.shape, device="cuda", generator=g) * 1e-2).to(torch.bfloat16) for _, p in used]
    twin = [p.detach().clone().requires_grad_(True) for _, p in named]
    ref2 = torch.optim.AdamW(twin, lr=3e-4, weight_decay=5e-5)
    ref2.load_state_dict(sd)
    twin_used = [t for (k, _), t in zip(named, twin) if not k.startswith("mlp_mixer.token_mix")]
    for t, gr in zip(twin_used, grads):
        t.grad = gr.clone()
    ref2.step()
    opt.zero_grad(set_to_none=False)
    for (_, p), gr in zip(used, grads):
        p.grad.copy_(gr)
    opt.step()
    assert opt.step_count == 3
    for (k, p), t in zip(used, twin_used):
        d = (p.float() - t.float()).abs().max()
        assert d <= 2.0 ** -7 * t.float().abs().max() + 1e-6, (k, float(d))      # one bf16 ulp: master vs in-place rounding


def test_predictor_with_hidden_layers_trains_with_dropout():
    """MultiLayerPredictor(n_layers = 4) in training mode (vit.py:310-318: two more Linear / GELU / Dropout(0.5) groups
    after the factorised layer): every GELU + Dropout pair is one fused pass; against fp32 math with the same masks."""
    from sfcvit import ops
    from sfcvit.models.vit import MultiLayerPredictor
    torch.manual_seed(0)
    B, N, D = 8, 16, 64
    head = MultiLayerPredictor(D, N, n_layers=4, num_classes=10).to("cuda", dtype=torch.bfloat16).train()
    x = torch.randn(B, N, D, device="cuda").to(torch.bfloat16).requires_grad_(True)
    torch.manual_seed(5)
    seeds = [ops.next_seed() for _ in range(3)]
    torch.manual_seed(5)
    y = head(x)
    y.float().sum().backward()
    assert y.shape == (B, 10) and torch.isfinite(y.float()).all()
    assert all(p.grad is not None and torch.isfinite(p.grad.float()).all() for p in head.parameters())
    # fp32 reference with the same masks
    P = {k: v.detach().float() for k, v in head.state_dict().items()}
    z = torch.nn.functional.layer_norm(x.detach().float(), (D,), P["0.weight"], P["0.bias"])
    h = torch.einsum("bnd,rd->bnr", z, P["1.W_emb"])
    t = torch.einsum("bnr,onr->bo", h, P["1.W_seq"])
    t = torch.nn.functional.gelu(t) * ops.dropout_mask(B, t.shape[1], 0.5, seeds[0]).float()
    for i, (lin, sd_) in enumerate(((4, seeds[1]), (7, seeds[2]))):
        t = t @ P[f"{lin}.weight"].t() + P[f"{lin}.bias"]
        t = torch.nn.functional.gelu(t) * ops.dropout_mask(B, t.shape[1], 0.5, sd_).float()
    want = t @ P["10.weight"].t() + P["10.bias"]
    assert (y.float() - want).abs().max() <= 3e-2 * want.abs().max() + 1e-3
    head.eval()
    with torch.no_grad():
        e1, e2 = head(x), head(x)
    assert torch.equal(e1, e2) and not torch.equal(e1, y.detach())


def test_torch_compile_traces_the_model_into_one_graph_and_matches_eager():
    """main.py:284: `model = torch.compile(model, mode="reduce-overhead")`.  The blocks of a VisionTransformer1D are
    `sfcvit::` custom ops with fake kernels and autograd formulas (sfcvit/library.py): Dynamo must trace the model into
    ONE graph with no graph breaks; the compiled module gives the eager logits bit for bit (eval), its backward the eager
    gradients (same kernels, fresh tensors instead of views of the flat gradient buffer), and mode="reduce-overhead"
    replays bit-identically."""
    import sfcvit.functional as F
    import sfcvit.library  # noqa: F401  (registers the ops)
    cfg, batch = MODEL_CASES["hilbert32_1d"]
    model = build_model(cfg)
    load_formula(model, cfg)
    model = model.to("cuda", dtype=torch.bfloat16).eval()
    x = formula.image_batch(batch, 3, cfg.img_size, cfg.img_size).cuda()
    tgt = formula.soft_targets(batch, cfg.num_classes).cuda()
    for name in ("patch_embed", "mixer_block", "encoder_layer", "predictor_head", "soft_ce"):
        assert hasattr(torch.ops.sfcvit, name) and hasattr(torch.ops.sfcvit, name + "_bwd" if name != "soft_ce" else name)
    with torch.no_grad():
        want = model(x)
    ex = torch._dynamo.explain(model)(x)
    assert ex.graph_break_count == 0 and ex.graph_count == 1, (ex.graph_break_count, ex.graph_count, ex.break_reasons)
    torch._dynamo.reset()

    # eager gradients (plain autograd: no optimizer, so no gradient slots on either side)
    loss_e = F.soft_target_cross_entropy(model(x), tgt)
    loss_e.backward()
    grads_e = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():
        p.grad = None

    def lossfn(m, xb, tb):
        return F.soft_target_cross_entropy(m(xb), tb)

    compiled = torch.compile(model)
    with torch.no_grad():
        assert torch.equal(compiled(x), want)
    assert all(k.startswith("_orig_mod.") for k in compiled.state_dict())      # the key prefix the reference's checkpoints carry
    loss_c = torch.compile(lossfn)(model, x, tgt)
    assert torch.equal(loss_c.detach(), loss_e.detach())
    loss_c.backward()
    for k, p in model.named_parameters():
        if k.startswith("mlp_mixer.token_mix"):
            assert p.grad is None
            continue
        assert torch.equal(p.grad, grads_e[k]), k
    torch._dynamo.reset()

    ro = torch.compile(model, mode="reduce-overhead")                            # main.py:284 literally
    with torch.no_grad():
        for _ in range(4):                                                       # warm-up, capture, replays
            got = ro(x).clone()
            assert torch.equal(got, want)
    # training mode (dropout on) traces as well: seeds are drawn inside the ops
    torch._dynamo.reset()
    model.train()
    ex = torch._dynamo.explain(model)(x)
    assert ex.graph_break_count == 0, ex.break_reasons
    torch.manual_seed(3)
    a = torch.compile(model)(x)
    torch.manual_seed(3)
    b = model(x)
    assert torch.equal(a, b)                                                     # same seeds, same masks as eager


def test_main_py_trains_on_a_caller_supplied_dataloader(tmp_path):
    """`main.py --data-module pkg:fn` takes the caller's loaders where the reference builds its torchvision
    pipeline inline (/root/reference/main.py:169-230): one epoch on tests/data_module_example.py's CPU DataLoaders,
    the reference's progress line (main.py:331-335) and a checkpoint with the reference's keys (main.py:345-354)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, os.path.join(root, "space-filling-curves-for-vision-transformers_amd", "main.py"), "--epochs", "1",
           "--batch-size", "32", "--tokenizer", "hilbert", "--embed-dim", "64", "--depth", "2", "--heads", "2", "--mlp-dim", "128",
           "--data-module", "data_module_example:loaders", "--checkpoint-dir", str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Epoch 1/1 | Train Loss:" in out.stdout and "synthetic" not in out.stdout
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint_hilbert.pt"), map_location="cpu", weights_only=True)
    assert {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "train_loss", "test_acc"} <= set(ck)
