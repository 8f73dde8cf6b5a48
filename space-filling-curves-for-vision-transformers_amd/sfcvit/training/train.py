"""Training step with the semantics of the reference's hot loop
(src/training/train.py:143-173): zero_grad -> forward -> soft-target CE -> backward ->
clip 1.0 -> AdamW -> scheduler, without the per-step host syncs (`.item()` x2 and the
per-tensor norm loop) and with optional data parallelism."""
import torch
import torch.nn as nn

from .. import functional as F


class SoftTargetCrossEntropy(nn.Module):
    """main.py:45-51."""

    def forward(self, inputs, targets):
        return F.soft_target_cross_entropy(inputs, targets)


def mixup_soft_targets(labels, num_classes, lam=0.7):
    """Deterministic stand-in for train.py:148-160: fixed-lambda MixUp of the one-hot labels with
    the batch rolled by one (SURVEY.md §8d synthetic-input definition)."""
    one = torch.nn.functional.one_hot(labels, num_classes).float()
    return lam * one + (1.0 - lam) * one.roll(1, 0)


def train_step(model, images, soft_targets, optimizer, scheduler=None, reducer=None):
    """One optimisation step; returns the (device, un-synchronised) loss."""
    optimizer.zero_grad()
    if reducer is not None:
        reducer.begin_step()
    logits = model(images)
    loss = F.soft_target_cross_entropy(logits, soft_targets)
    loss.backward()
    if reducer is not None:
        reducer.finish()                 # all buckets reduced (SUM); optimizer.grad_scale = 1/world
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach()
