#!/usr/bin/env python3
"""Entry script with the structure of the reference's main.py (seeds, tokenizer menu, model, AdamW +
warm-up cosine schedule, MixUp/CutMix epochs, evaluate, checkpoint on best accuracy) running on the
MI355X kernels.  The reference hard-codes everything and needs torchvision + a CIFAR-10 download; here
the same defaults are arguments and `--synthetic` (default when torchvision is missing) feeds random
CIFAR-shaped batches so that the script runs on an air-gapped GPU box.

    python main.py --epochs 2 --synthetic                       # reference default model on 32x32
    python main.py --tokenizer hilbert --img-size 224 --patch-size 256 --embed-dim 768 --depth 12 \\
                   --heads 12 --mlp-dim 3072 --classes 1000 --batch-size 256 --synthetic
    torchrun --nproc-per-node 8 main.py ...                     # data parallel (one process per GPU)
"""
import argparse
import os
import sys

# dmabuf IPC is the only kind this pool's host driver supports: without it RCCL (and any cross-process sharing of
# device memory) fails with "hipIpcGetMemHandle: invalid argument".  Must be in the environment before HIP initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sfcvit.models.vit import VisionTransformer1D                                  # noqa: E402
import sfcvit.tokenizers as T                                                       # noqa: E402
from sfcvit.training import FusedAdamW, GradReducer, GraphedTrainStep, SoftTargetCrossEntropy, WarmupCosine   # noqa: E402
from sfcvit.training.loops import evaluate, train_with_mixup_or_cutmix             # noqa: E402


class SyntheticLoader:
    """CIFAR-shaped random batches, regenerated deterministically every epoch."""

    def __init__(self, n, batch, img, classes, seed, device):
        self.n, self.batch, self.img, self.classes, self.seed, self.device = n, batch, img, classes, seed, device
        self.dataset = range(n)

    def __len__(self):
        return self.n // self.batch

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed)
        for _ in range(len(self)):
            yield (torch.randn(self.batch, 3, self.img, self.img, device=self.device, generator=g),
                   torch.randint(0, self.classes, (self.batch,), device=self.device, generator=g))


def build_tokenizer(a):
    one_d = {"raster": T.RasterScan1DEmbedding, "hilbert": T.HilbertEmbedding1D, "peano": T.PeanoEmbedding1D,
             "moore": T.MooreEmbedding1D, "onion": T.OnionEmbedding1D, "morton": T.MortonEmbedding1D,
             "zigzag": T.ZigzagEmbedding, "hilbert2d": T.HilbertEmbedding}           # main.py:232-240 (+ _2D/hilbert)
    hier = {"hier_raster": T.HierarchicalRasterScanEmbedding, "hier_hilbert": T.HierarchicalHilbertEmbedding,
            "hier_peano": T.HierarchicalPeanoEmbedding, "hier_moore": T.HierarchicalMooreEmbedding,
            "hier_onion": T.HierarchicalOnionEmbedding, "hier_morton": T.HierarchicalMortonEmbedding}   # main.py:242-250
    if a.tokenizer in one_d:
        return one_d[a.tokenizer](a.img_size, a.patch_size, 3, a.embed_dim)
    if a.tokenizer not in hier:
        raise SystemExit(f"--tokenizer must be one of {sorted(one_d) + sorted(hier)}")
    return hier[a.tokenizer](a.img_size, 3, a.levels, a.embed_dim)          # main.py:269-274: ([16, 4, 1], 256)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokenizer", default="hier_morton")       # the reference's active menu entry (main.py:232-240)
    ap.add_argument("--img-size", type=int, default=32)
    ap.add_argument("--patch-size", type=int, default=256)
    ap.add_argument("--levels", type=int, nargs="+", default=[16, 4, 1])
    ap.add_argument("--embed-dim", type=int, default=256)
    ap.add_argument("--depth", type=int, default=8)             # main.py:276-282
    ap.add_argument("--heads", type=int, default=4)             # head dim 768 / 4 = 192 (attention_wide.hip)
    ap.add_argument("--mlp-dim", type=int, default=512)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=512)      # main.py:228
    ap.add_argument("--epochs", type=int, default=300)          # main.py:306
    ap.add_argument("--warmup-epochs", type=int, default=10)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--weight-decay", type=float, default=5e-5)
    ap.add_argument("--train-size", type=int, default=50000)
    ap.add_argument("--test-size", type=int, default=10000)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--data-module", default=None,
                    help="pkg.module:function returning (train_loader, test_loader) -- the caller's own DataLoaders in place "
                         "of the reference's inline CIFAR-10 pipeline (main.py:169-230)")
    ap.add_argument("--checkpoint-dir", default="./vit_checkpoints")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--resume-model-only", action="store_true",
                    help="take only the weights from --resume (e.g. a checkpoint the REFERENCE's main.py wrote: its "
                         "optimizer / scheduler states are torch.optim formats, see INTEGRATION.md)")
    ap.add_argument("--graph", action="store_true",
                    help="replay every training step from one hipGraph (the reference's torch.compile(model, "
                         "mode='reduce-overhead'), main.py:284, for the whole step); single process")
    a = ap.parse_args()

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        from sfcvit.training.distributed import dist_timeout
        dist.init_process_group("nccl", device_id=device, timeout=dist_timeout())      # bounded collectives (SFCVIT_DIST_TIMEOUT)
    seed = 42                                                   # main.py:151-154
    torch.manual_seed(seed)
    np.random.seed(seed + rank)

    try:
        import torchvision  # noqa: F401
        have_tv = True
    except ImportError:
        have_tv = False
    per_rank = a.batch_size // world
    if a.data_module:
        # The reference builds its loaders inline (main.py:169-230: torchvision CIFAR-10 + v2 transforms, DataLoader with
        # 16 workers).  That pipeline is the caller's; this is the hook for it: `pkg.mod:fn` names a factory
        #     fn(batch_size=, img_size=, classes=, rank=, world=, seed=) -> (train_loader, test_loader)
        # returning iterables of (images [B, 3, H, W] float, labels [B] int64) with __len__, on any device (the loops move
        # every batch to the GPU, src/training/train.py:144-145), each rank's own shard, every batch of `batch_size` rows.
        import importlib
        mod, _, fn = a.data_module.partition(":")
        if not mod or not fn:
            raise SystemExit("--data-module takes pkg.module:function")
        factory = getattr(importlib.import_module(mod), fn)
        train_loader, test_loader = factory(batch_size=per_rank, img_size=a.img_size, classes=a.classes, rank=rank,
                                            world=world, seed=seed)
    else:
        if not a.synthetic:
            print("no --data-module given (the reference's torchvision CIFAR-10 pipeline, main.py:169-230, is not part of "
                  "this package" + ("" if have_tv else " and torchvision is not installed") + "): using --synthetic data")
            a.synthetic = True
        train_loader = SyntheticLoader(a.train_size // world, per_rank, a.img_size, a.classes, seed + 1 + rank, device)
        test_loader = SyntheticLoader(a.test_size // world, per_rank, a.img_size, a.classes, seed + 1001 + rank, device)

    patch_embed = build_tokenizer(a)
    model = VisionTransformer1D(patch_embed=patch_embed, depth=a.depth, n_heads=a.heads, mlp_dim=a.mlp_dim,
                                num_classes=a.classes).to(device, dtype=torch.bfloat16)   # main.py:157: bf16 parameters
    train_criterion, test_criterion = SoftTargetCrossEntropy(), nn.CrossEntropyLoss()
    optimizer = FusedAdamW(model.parameters(), lr=a.lr, weight_decay=a.weight_decay, max_grad_norm=1.0)
    reducer = GradReducer(optimizer, overlap=not a.graph) if world > 1 else None    # --graph: collectives between two graphs
    scheduler = WarmupCosine(optimizer, a.warmup_epochs * len(train_loader), a.epochs * len(train_loader))
    start_epoch, best = 0, 0.0
    if a.resume:
        ck = torch.load(a.resume, map_location=device, weights_only=True)
        # the reference saves the torch.compile wrapper's state_dict: same keys behind an "_orig_mod." prefix
        msd = {(k[len("_orig_mod."):] if k.startswith("_orig_mod.") else k): v for k, v in ck["model_state_dict"].items()}
        model.load_state_dict(msd)
        osd = ck.get("optimizer_state_dict") or {}
        if osd and not a.resume_model_only:
            # this repository's flat fp32 state (FusedAdamW.state_dict), or a torch.optim.AdamW state_dict as the reference's
            # checkpoints hold (main.py:345-354): Adam moments and step count are mapped into the flat buffers
            optimizer.load_state_dict(osd)
            start_epoch, best = ck["epoch"] + 1, ck.get("test_acc", 0.0)
            ssd = ck.get("scheduler_state_dict") or {}
            # ours: {"n": steps taken}; the reference's LambdaLR (transformers' cosine schedule): {"last_epoch": steps taken}
            scheduler.n = ssd.get("n", ssd.get("last_epoch", start_epoch * len(train_loader)))
            optimizer.lr = scheduler.lr_at(scheduler.n)
    os.makedirs(a.checkpoint_dir, exist_ok=True)
    ckpt = os.path.join(a.checkpoint_dir, f"checkpoint_{a.tokenizer}.pt")
    graphed = None
    if a.graph:
        model.train()
        graphed = GraphedTrainStep(model, torch.zeros(per_rank, 3, a.img_size, a.img_size, device=device),
                                   torch.zeros(per_rank, a.classes, device=device), optimizer, scheduler, reducer=reducer)

    for epoch in range(start_epoch, a.epochs):
        tr_loss, tr_acc = train_with_mixup_or_cutmix(model, train_loader, train_criterion, optimizer, scheduler,
                                                     device, reducer=reducer, graphed=graphed)
        te_loss, te_acc = evaluate(model, test_loader, test_criterion, device)
        if world > 1:                                  # equal shards per rank: the global figures are the rank means
            t = torch.tensor([tr_loss, tr_acc, te_loss, te_acc], device=device, dtype=torch.float64)
            dist.all_reduce(t)
            tr_loss, tr_acc, te_loss, te_acc = (t / world).tolist()
        if rank == 0:
            print(f"Epoch {epoch + 1}/{a.epochs} | Train Loss: {tr_loss:.4f}, Train Acc: {tr_acc:.4f} | "
                  f"Test Loss: {te_loss:.4f}, Test Acc: {te_acc:.4f}")            # main.py:331-335
            if te_acc > best or epoch == start_epoch:
                best = te_acc
                torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                            "optimizer_state_dict": optimizer.state_dict() if optimizer.master is not None else {},
                            "scheduler_state_dict": {"n": scheduler.n}, "train_loss": tr_loss, "train_acc": tr_acc,
                            "test_loss": te_loss, "test_acc": te_acc}, ckpt)      # main.py:345-354 keys
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
