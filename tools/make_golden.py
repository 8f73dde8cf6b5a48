#!/usr/bin/env python3
"""Generate tests/golden/* by importing the reference (authoring container only).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_golden.py

The reference (/root/reference, read-only) never travels to the GPU box; what is
committed is data only: curve tables / hashes, and outputs (logits, loss, sampled
gradients) of the reference's own classes evaluated on formula-generated inputs
and weights (oracle/formula.py).  Without /root/reference this script does nothing.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def sample_idx(numel, k=16):
    if numel <= k:
        return list(range(numel))
    return [int(i) for i in np.linspace(0, numel - 1, k).astype(np.int64)]


def main():
    if not os.path.isdir(REF):
        print("reference not present: fixtures left as committed")
        return 0
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    from src.curves import space_filling_curves as rc
    from src.tokenizers._1D.hilbert_embedding1D import HilbertEmbedding1D
    from src.tokenizers._1D.morton_embedding1D import MortonEmbedding1D
    from src.tokenizers._1D.zigzag_embedding1D import RasterScan1DEmbedding
    from src.tokenizers.multiscale.multi_hilbert import SFCEmbedding1D
    from src.models.vit import VisionTransformer, VisionTransformer1D
    from oracle import formula
    from oracle.cases import MODEL_CASES, CURVE_SMALL_N, CURVE_SHA_N, CURVE_KINDS, HIER_CASES
    from src.tokenizers.multiscale.multi_hilbert import HierarchicalHilbertEmbedding
    from src.tokenizers.multiscale.multi_morton import HierarchicalMortonEmbedding

    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    fns = {"hilbert": rc.hilbert_curve, "z": rc.z_curve, "moore": rc.moore_curve,
           "peano": rc.peano_curve}

    # ---- (1) curve tables ------------------------------------------------
    small, sha = {}, {}
    for kind in CURVE_KINDS:
        for n in CURVE_SMALL_N:
            ij = np.array(rc.embed_and_prune_sfc(fns[kind], n, n), dtype=np.int64)
            flat = (ij[:, 0] * n + ij[:, 1]).astype(np.int32)
            small[f"{kind}_{n}"] = flat
            sha[f"{kind}_{n}"] = hashlib.sha256(flat.tobytes()).hexdigest()
    for kind in ("hilbert", "z"):
        for n in CURVE_SHA_N:
            ij = np.array(rc.embed_and_prune_sfc(fns[kind], n, n), dtype=np.int64)
            flat = (ij[:, 0] * n + ij[:, 1]).astype(np.int32)
            sha[f"{kind}_{n}"] = hashlib.sha256(flat.tobytes()).hexdigest()
            small[f"{kind}_{n}_head"] = flat[:64]
            small[f"{kind}_{n}_tail"] = flat[-64:]
    np.savez_compressed(os.path.join(GOLD, "curves_small.npz"), **small)
    with open(os.path.join(GOLD, "curves_sha.json"), "w") as f:
        json.dump(sha, f, indent=1, sort_keys=True)

    # ---- (2)-(5) models ----------------------------------------------------
    curve_fn = {"hilbert": rc.hilbert_curve, "z": rc.z_curve}
    manifest = {}
    for name, (cfg, batch) in MODEL_CASES.items():
        if cfg.tokenizer == "hilbert1d":
            pe = HilbertEmbedding1D(cfg.img_size, cfg.patch_size, cfg.in_channels, cfg.embed_dim)
        elif cfg.tokenizer == "morton1d":
            pe = MortonEmbedding1D(cfg.img_size, cfg.patch_size, cfg.in_channels, cfg.embed_dim)
        elif cfg.tokenizer == "raster1d":
            pe = RasterScan1DEmbedding(cfg.img_size, cfg.patch_size, cfg.in_channels, cfg.embed_dim)
        else:
            pe = SFCEmbedding1D(cfg.img_size, cfg.pre_patch_size, cfg.patch_size, cfg.in_channels,
                                cfg.embed_dim, curve_fn[cfg.curve])
            pe.n_patches = pe.n_final_patches      # the attribute vit.py:354,422 reads
        cls = VisionTransformer1D if cfg.variant == "1d" else VisionTransformer
        model = cls(pe, depth=cfg.depth, n_heads=cfg.n_heads, mlp_dim=cfg.mlp_dim,
                    num_classes=cfg.num_classes)
        sd = model.state_dict()
        manifest[name] = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}
        model.load_state_dict(formula.fill_state_dict(sd))
        model.eval()                                   # dropout off: parity is eval-mode
        x = formula.image_batch(batch, cfg.in_channels, cfg.img_size, cfg.img_size)
        tgt = formula.soft_targets(batch, cfg.num_classes)
        tokens = model.patch_embed(x)
        logits = model(x)
        logp = torch.log_softmax(logits, dim=-1)
        loss = -(tgt * logp).sum(-1).mean()            # main.py:49-51
        loss.backward()
        out = {
            "batch": batch,
            "logits": logits.detach().double().tolist(),
            "loss": float(loss.detach()),
            "tokens_sample_idx": sample_idx(tokens.numel(), 64),
            "tokens_l2": float(tokens.detach().double().norm()),
            "grads": {},
        }
        tf = tokens.detach().flatten()
        out["tokens_sample"] = [float(tf[i]) for i in out["tokens_sample_idx"]]
        for k, p in model.named_parameters():
            if p.grad is None:
                out["grads"][k] = None
                continue
            g = p.grad.detach().flatten().double()
            idx = sample_idx(g.numel())
            out["grads"][k] = {"l2": float(g.norm()), "idx": idx, "val": [float(g[i]) for i in idx]}
        with open(os.path.join(GOLD, f"model_{name}.json"), "w") as f:
            json.dump(out, f)
        print(name, "loss", out["loss"], "logits[0][:3]", out["logits"][0][:3])
    # ---- hierarchical tokenizers ----------------------------------------------
    hier = {}
    for name, (img, cin, plist, dim, curve, batch) in HIER_CASES.items():
        cls = HierarchicalMortonEmbedding if curve == "z" else HierarchicalHilbertEmbedding
        mod = cls(img, cin, plist, dim)
        mod.load_state_dict(formula.fill_state_dict(mod.state_dict()))
        x = formula.image_batch(batch, cin, img, img)
        y = mod(x).detach()
        idx = sample_idx(y.numel(), 64)
        yf = y.flatten()
        hier[name] = {"shape": list(y.shape), "l2": float(y.double().norm()), "idx": idx,
                      "val": [float(yf[i]) for i in idx], "n_patches": mod.n_patches, "embed_dim": mod.embed_dim,
                      "keys": {k: list(v.shape) for k, v in mod.state_dict().items()}}
        print(name, hier[name]["shape"], hier[name]["l2"])
    with open(os.path.join(GOLD, "hierarchical.json"), "w") as f:
        json.dump(hier, f)
    # ---- remaining tokenizers ------------------------
    import importlib
    from oracle.cases import TOKENIZER_CASES, RANDPERM_SEED, SPIRAL_N, HILBERT_T_N
    toks = {}
    for name, (modname, clsname, args, kind, batch) in TOKENIZER_CASES.items():
        mod = getattr(importlib.import_module(modname), clsname)(*args)
        mod.load_state_dict(formula.fill_state_dict(mod.state_dict()))
        x = formula.image_batch(batch, 3, args[0], args[0])
        torch.manual_seed(RANDPERM_SEED)
        y = mod(x).detach()
        idx = sample_idx(y.numel(), 64)
        yf = y.flatten()
        toks[name] = {"shape": list(y.shape), "l2": float(y.double().norm()), "idx": idx,
                      "val": [float(yf[i]) for i in idx], "n_patches": int(getattr(mod, "n_patches", getattr(mod, "n_final_patches", -1))),
                      "embed_dim": int(getattr(mod, "embed_dim", y.shape[-1])),
                      "keys": {k: list(v.shape) for k, v in mod.state_dict().items()}}
        print(name, toks[name]["shape"], toks[name]["l2"])
    with open(os.path.join(GOLD, "tokenizers.json"), "w") as f:
        json.dump(toks, f)
    from src.tokenizers._1D.onion_embedding1D import OnionEmbedding1D
    from src.tokenizers._2D.hilbert_embedding import HilbertEmbedding
    extra = {}
    for n in SPIRAL_N:
        r, c = OnionEmbedding1D(n, 1, 3, 4).onion_indices(n, n)
        extra[f"spiral_{n}"] = (np.asarray(r) * n + np.asarray(c)).astype(np.int32)
    for n in HILBERT_T_N:
        extra[f"hilbert_t_{n}"] = HilbertEmbedding(n, 1, 3, 4).hilbert_indices.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(GOLD, "curves_extra.npz"), **extra)
    # ---- altvit (SimpleViT / HilbertViT) --------------------------------------------
    from oracle.cases import ALTVIT_CASES
    from src.models import altvit as ref_altvit
    alt = {}
    for name, (clsname, kw, batch) in ALTVIT_CASES.items():
        mod = getattr(ref_altvit, clsname)(**kw).eval()
        sd0 = mod.state_dict()
        filled = formula.fill_state_dict(sd0)
        filled["pos_embedding"] = sd0["pos_embedding"]           # computed buffer, not a weight
        mod.load_state_dict(filled)
        x = formula.image_batch(batch, 3, kw["image_size"], kw["image_size"])
        tgt = formula.soft_targets(batch, kw["num_classes"])
        logits = mod(x)
        loss = torch.sum(-tgt * torch.nn.functional.log_softmax(logits, dim=-1), dim=-1).mean()
        loss.backward()
        grads = {k: float(p.grad.double().norm()) for k, p in mod.named_parameters()}
        alt[name] = {"logits": logits.detach().tolist(), "loss": float(loss), "grad_norm": grads,
                     "pos_embedding_l2": float(sd0["pos_embedding"].double().norm()),
                     "pos_embedding_head": sd0["pos_embedding"].flatten()[:8].tolist(),
                     "keys": {k: list(v.shape) for k, v in sd0.items()}}
        print(name, "loss", float(loss))
    with open(os.path.join(GOLD, "altvit.json"), "w") as f:
        json.dump(alt, f)
    with open(os.path.join(GOLD, "state_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
