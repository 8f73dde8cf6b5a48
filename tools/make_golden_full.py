#!/usr/bin/env python3
"""Fixtures at the BASELINE configurations' true model dimensions + optimisation-step fixtures
(authoring container only; imports the reference from /root/reference, which never travels).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_golden_full.py

Writes tests/golden/full_<case>.json (oracle/cases.py FULL_CASES: eval logits, loss, per-parameter gradient
L2 norms + 8 sampled gradient values of the reference's own VisionTransformer1D on formula weights) and
tests/golden/train_<case>.json (TRAIN_CASES: per-step loss, total gradient norm, and per-parameter
L2 norm + samples of the weights after N steps of zero_grad -> forward -> soft-target CE -> backward ->
clip_grad_norm_(1.0, foreach=False) -> AdamW, i.e. src/training/train.py:153-167 with main.py:288-289's optimizer).
Data only; without /root/reference this script does nothing.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def sample_idx(numel, k=8):
    if numel <= k:
        return list(range(numel))
    return [int(i) for i in np.linspace(0, numel - 1, k).astype(np.int64)]


def build(cfg):
    from src.tokenizers._1D.hilbert_embedding1D import HilbertEmbedding1D
    from src.tokenizers._1D.morton_embedding1D import MortonEmbedding1D
    from src.tokenizers._1D.zigzag_embedding1D import RasterScan1DEmbedding
    from src.models.vit import VisionTransformer, VisionTransformer1D
    cls = {"hilbert1d": HilbertEmbedding1D, "morton1d": MortonEmbedding1D, "raster1d": RasterScan1DEmbedding}[cfg.tokenizer]
    pe = cls(cfg.img_size, cfg.patch_size, cfg.in_channels, cfg.embed_dim)
    mcls = VisionTransformer1D if cfg.variant == "1d" else VisionTransformer
    return mcls(pe, depth=cfg.depth, n_heads=cfg.n_heads, mlp_dim=cfg.mlp_dim, num_classes=cfg.num_classes)


def main():
    if not os.path.isdir(REF):
        print("reference not present: fixtures left as committed")
        return 0
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    from oracle import formula
    from oracle.cases import FULL_CASES, MODEL_CASES, TRAIN_CASES
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    for name, (cfg, batch) in FULL_CASES.items():
        if only and name not in only:
            continue
        torch.manual_seed(0)
        model = build(cfg)
        model.load_state_dict(formula.fill_state_dict(model.state_dict()))
        model.eval()
        x = formula.image_batch(batch, cfg.in_channels, cfg.img_size, cfg.img_size)
        tgt = formula.soft_targets(batch, cfg.num_classes)
        logits = model(x)
        loss = -(tgt * torch.log_softmax(logits, dim=-1)).sum(-1).mean()       # main.py:49-51
        loss.backward()
        out = {"batch": batch, "loss": float(loss.detach()), "grads": {}}
        if logits.numel() > 8192:          # large batches: 40 evenly spaced class columns of every row (keeps the fixture small)
            cols = sample_idx(logits.shape[1], 40)
            out["logit_cols"] = cols
            out["logits"] = logits.detach()[:, cols].double().tolist()
        else:
            out["logits"] = logits.detach().double().tolist()
        for k, p in model.named_parameters():
            if p.grad is None:
                out["grads"][k] = None
                continue
            g = p.grad.detach().flatten().double()
            idx = sample_idx(g.numel())
            out["grads"][k] = {"l2": float(g.norm()), "idx": idx, "val": [float(g[i]) for i in idx]}
        with open(os.path.join(GOLD, f"full_{name}.json"), "w") as f:
            json.dump(out, f)
        print(name, "loss", out["loss"], "max|logit|", float(logits.abs().max()), flush=True)
    for name, (case, steps, lr, wd) in TRAIN_CASES.items():
        if only and ("train_" + name) not in only:
            continue
        cfg, batch = MODEL_CASES[case]
        model = build(cfg)
        model.load_state_dict(formula.fill_state_dict(model.state_dict()))
        model.eval()                                       # dropout off; gradients flow as in train()
        x = formula.image_batch(batch, cfg.in_channels, cfg.img_size, cfg.img_size)
        tgt = formula.soft_targets(batch, cfg.num_classes)
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)   # main.py:288-289 (rates per case)
        losses, norms = [], []
        for _ in range(steps):
            opt.zero_grad()                                # train.py:153
            logits = model(x)                              # :156
            loss = torch.sum(-tgt * torch.nn.functional.log_softmax(logits, dim=-1), dim=-1).mean()   # main.py:49-51
            loss.backward()                                # train.py:163
            norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, foreach=False)))  # :165
            opt.step()                                     # :166
            losses.append(float(loss.detach()))
        out = {"case": case, "steps": steps, "lr": lr, "weight_decay": wd, "loss": losses, "grad_norm": norms, "params": {}}
        for k, p in model.named_parameters():
            v = p.detach().flatten().double()
            idx = sample_idx(v.numel())
            out["params"][k] = {"l2": float(v.norm()), "idx": idx, "val": [float(v[i]) for i in idx]}
        with open(os.path.join(GOLD, f"train_{name}.json"), "w") as f:
            json.dump(out, f)
        print("train", name, losses, norms, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
