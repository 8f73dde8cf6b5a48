#!/usr/bin/env python3
"""Regenerate tests/golden/hierarchical.json only (same recipe as tools/make_golden.py's hierarchical section; run in
the authoring container, where /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_golden_hier.py

Outputs of the reference's Hierarchical{Morton,Hilbert}Embedding on formula-generated weights and images
(oracle/formula.py) for every entry of oracle.cases.HIER_CASES: data only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    if not os.path.isdir(REF):
        print("reference not present: fixture left as committed")
        return 0
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    from make_golden import GOLD, sample_idx
    from oracle import formula
    from oracle.cases import HIER_CASES
    from src.tokenizers.multiscale.multi_hilbert import HierarchicalHilbertEmbedding
    from src.tokenizers.multiscale.multi_morton import HierarchicalMortonEmbedding
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
    return 0


if __name__ == "__main__":
    sys.exit(main())
