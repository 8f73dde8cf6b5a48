"""End-to-end parity of the HIP path (through the C ABI) with the oracle and with the
fixtures generated from the reference itself.

Stated tolerance (north star: "matching logits within a stated fp tolerance"): the HIP
path computes in bf16 with fp32 accumulation, the oracle / reference in fp32, so
   max|logit_hip - logit_ref| <= 3e-2 * max|logit_ref|            (eval mode)
   |loss_hip - loss_ref|      <= 2e-3 * |loss_ref| + 2e-3
   per-parameter gradient: cosine >= 0.99 and |‖g_hip‖/‖g_ref‖ - 1| <= 5e-2 (bf16 grads)
Curve index buffers must be identical (bit-exact)."""
import json
import os

import pytest
import torch

from oracle import formula, vit_oracle
from oracle.cases import MODEL_CASES
from test_host_cpu import build_model

pytestmark = pytest.mark.gpu

LOGIT_TOL = 3e-2


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
