"""Epoch loops with the semantics of src/training/train.py (MixUp / CutMix, soft-target CE, clip 1.0,
optimizer + scheduler step per batch; evaluate = eval-mode forward + CE + accuracy), on the HIP modules.

Differences that do not change the math: no torch.autocast (the modules compute in bf16 with fp32
accumulation themselves), running loss / accuracy stay on the device and are read once per epoch
(the reference calls .item() twice per step, train.py:170-172), and FusedAdamW.step() does the
clip_grad_norm_(1.0) inside the optimizer kernel (train.py:165-166)."""
import numpy as np
import torch
import torch.nn.functional as TF

from .optim import FusedAdamW


def mixup_data(x, y, alpha=0.2):
    """train.py:7-14."""
    lam = np.random.beta(alpha, alpha) if alpha > 0 else 1.0
    idx = torch.randperm(x.size(0), device=x.device)
    return lam * x + (1 - lam) * x[idx], y, y[idx], lam


def rand_bbox(H, W, lam):
    """train.py:17-30."""
    cut_rat = np.sqrt(1.0 - lam)
    cut_w, cut_h = int(W * cut_rat), int(H * cut_rat)
    cx, cy = np.random.randint(W), np.random.randint(H)
    return (np.clip(cx - cut_w // 2, 0, W), np.clip(cy - cut_h // 2, 0, H),
            np.clip(cx + cut_w // 2, 0, W), np.clip(cy + cut_h // 2, 0, H))


def cutmix_data(x, y, alpha=0.2):
    """train.py:33-47 (mutates x in place, as the reference does)."""
    lam = np.random.beta(alpha, alpha) if alpha > 0 else 1.0
    _, _, H, W = x.size()
    idx = torch.randperm(x.size(0), device=x.device)
    bbx1, bby1, bbx2, bby2 = rand_bbox(H, W, lam)
    x[:, :, bbx1:bbx2, bby1:bby2] = x[idx, :, bbx1:bbx2, bby1:bby2]
    lam = 1 - ((bbx2 - bbx1) * (bby2 - bby1) / (H * W))
    return x, y, y[idx], lam


def mixup_criterion(criterion, pred, y_a, y_b, lam):
    """train.py:50-54."""
    return lam * criterion(pred, y_a) + (1 - lam) * criterion(pred, y_b)


def _plain_epoch(model, train_loader, criterion, optimizer, scheduler, device):
    model.train()
    total_loss = torch.zeros((), device=device)
    correct = torch.zeros((), device=device)
    for images, labels in train_loader:
        images, labels = images.to(device), labels.to(device)
        if hasattr(optimizer, "begin_step"):
            optimizer.begin_step()
        optimizer.zero_grad()
        outputs = model(images)
        loss = criterion(outputs.float(), labels)
        loss.backward()
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        total_loss += loss.detach().float() * images.size(0)
        correct += (outputs.argmax(dim=1) == labels).sum()
    n = len(train_loader.dataset)
    return float(total_loss) / n, float(correct) / n


def train(model, train_loader, criterion, optimizer, device):
    """train.py:57-77: hard labels, no scheduler.  The reference does not clip here: construct
    FusedAdamW(..., max_grad_norm=None) for the same behaviour."""
    return _plain_epoch(model, train_loader, criterion, optimizer, None, device)


def train_with_scheduler(model, train_loader, criterion, optimizer, scheduler, device):
    """train.py:102-130: as `train` with a per-step scheduler."""
    return _plain_epoch(model, train_loader, criterion, optimizer, scheduler, device)


def train_with_mixup_or_cutmix(model, train_loader, criterion, optimizer, scheduler, device,
                               mixup_alpha=0.2, cutmix_alpha=1.0, mix_prob=0.5, reducer=None, graphed=None):
    """train.py:133-178.  `optimizer` is a FusedAdamW (clip inside) or any torch optimizer.
    graphed: a sfcvit.training.GraphedTrainStep built on this model / optimizer / scheduler with static buffers of the
    loader's batch shape -- the step (forward, soft-target CE, backward, clip, AdamW, scheduler) is then one hipGraph
    replay per batch (what main.py:284's torch.compile(mode="reduce-overhead") is after); batches of another size run
    eagerly."""
    model.train()
    total_loss = torch.zeros((), device=device)
    total_correct = torch.zeros((), device=device)
    total_samples = 0
    for images, labels in train_loader:
        images, labels = images.to(device), labels.to(device)
        if np.random.rand() < mix_prob:
            images, y_a, y_b, lam = mixup_data(images, labels, alpha=mixup_alpha)
        else:
            images, y_a, y_b, lam = cutmix_data(images, labels, alpha=cutmix_alpha)
        if graphed is not None and images.shape == graphed.images.shape:
            num_classes = graphed.targets.size(1)
            graphed.images.copy_(images)
            graphed.targets.copy_(lam * TF.one_hot(y_a, num_classes).float() + (1 - lam) * TF.one_hot(y_b, num_classes).float())
            loss = graphed()
            outputs = graphed.logits
            preds = outputs.argmax(dim=1)
            total_correct += (lam * (preds == y_a).float() + (1 - lam) * (preds == y_b).float()).sum()
            total_loss += loss.float() * images.size(0)
            total_samples += images.size(0)
            continue
        if hasattr(optimizer, "begin_step"):
            optimizer.begin_step()       # a batch the graph cannot replay (odd size): the device step state advances here
        optimizer.zero_grad()
        if reducer is not None:
            reducer.begin_step()
        outputs = model(images)
        num_classes = outputs.size(1)
        soft_targets = lam * TF.one_hot(y_a, num_classes).float() + (1 - lam) * TF.one_hot(y_b, num_classes).float()
        loss = criterion(outputs, soft_targets)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        if not isinstance(optimizer, FusedAdamW):
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, foreach=False)
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        preds = outputs.argmax(dim=1)
        total_correct += (lam * (preds == y_a).float() + (1 - lam) * (preds == y_b).float()).sum()
        total_loss += loss.detach().float() * images.size(0)
        total_samples += images.size(0)
    return float(total_loss) / total_samples, float(total_correct) / total_samples


def evaluate(model, test_loader, criterion, device):
    """train.py:80-99."""
    model.eval()
    total_loss = torch.zeros((), device=device)
    correct = torch.zeros((), device=device)
    n = 0
    with torch.no_grad():
        for images, labels in test_loader:
            images, labels = images.to(device), labels.to(device)
            outputs = model(images)
            total_loss += criterion(outputs.float(), labels) * images.size(0)
            correct += (outputs.argmax(dim=1) == labels).sum()
            n += images.size(0)
    return float(total_loss) / n, float(correct) / n
