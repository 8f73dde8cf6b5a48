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


class GraphedTrainStep:
    """A whole training step (zero_grad -> forward -> soft-target CE -> backward -> clip -> AdamW) captured once into a
    hipGraph and replayed: what main.py:284's torch.compile(model, mode="reduce-overhead") is after, for the entire
    step.  ~700 launches per ViT-B step (430 for ViT-Tiny) cost 7-11 ms of Python + ctypes per step when issued one
    by one; a replay is one call.

    `images` / `targets` are STATIC buffers: copy each new batch into them (`.copy_`) before calling; `.logits` is the
    static output of the last replay (accuracy bookkeeping of the epoch loops).  Needs the
    optimizer's device-resident step state (dropout seeds and Adam's step count would otherwise be frozen into the
    graph as by-value kernel arguments).  Shapes, model mode (train / eval) and dropout rates are fixed at capture.
    Data parallelism: pass `reducer=GradReducer(optimizer, overlap=False)`; the step is then TWO graphs -- forward +
    backward, and clip + AdamW -- with the gradient all-reduce issued between them (collectives launched from autograd
    hooks cannot replay from a graph; the price is their overlap with backward, which a launch-bound model -- the case
    a graph is for -- does not miss).
    The warm-up steps before the capture are real optimisation steps on whatever the static buffers hold; with
    `preserve_state` (default) parameters, optimizer moments, step count and scheduler position are put back
    afterwards, so that a training script that switches to the graphed step trains exactly as before."""

    def __init__(self, model, images, targets, optimizer, scheduler=None, warmup=3, preserve_state=True, reducer=None):
        self.model, self.images, self.targets, self.opt, self.sched = model, images, targets, optimizer, scheduler
        self.logits = None
        self.reducer = reducer
        if reducer is not None and reducer.overlap:
            raise ValueError("GraphedTrainStep needs GradReducer(optimizer, overlap=False): hooks cannot run inside a graph replay")
        optimizer.use_device_state(images.device)
        snap = self._snapshot() if preserve_state else None
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                 # warm up off the default stream (allocator pools, flat buffers,
            for _ in range(max(1, warmup)):           # LDS attributes, tile-queue slots: nothing of that may happen in capture)
                self._step()                          # eager: optimizer.step() counts on the host itself
                self._after(replayed=False)
        cur.wait_stream(side)
        if snap is not None:
            self._restore(snap)
            if reducer is not None:
                # _restore put back THIS rank's pre-warm-up state, i.e. it undid the rank-0 broadcast of the reducer's
                # first step: a rank that was built or loaded differently would drift again
                reducer.sync_replicas()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        if reducer is None:
            with torch.cuda.graph(self.graph):        # records, does not run: device state and host mirror stay as they are
                self.loss = self._step()
        else:                                         # two graphs around the collectives; one memory pool
            with torch.cuda.graph(self.graph):
                self.loss = self._fwd_bwd()
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool()):
                self.opt.step()

    def _fwd_bwd(self):
        self.opt.begin_step()
        self.opt.zero_grad()
        logits = self.model(self.images)
        loss = F.soft_target_cross_entropy(logits, self.targets)
        loss.backward()
        if self.reducer is not None and self.opt.flat_grad is not None:
            self.opt.adopt_all()                      # every gradient in the flat buffer before the collectives read it
        self.logits = logits.detach()
        return loss.detach()

    def _step(self):                                  # train_step() below, keeping the logits
        loss = self._fwd_bwd()
        if self.reducer is not None:
            self.reducer.begin_step()
            self.reducer.finish()
        self.opt.step()
        return loss

    def _snapshot(self):
        o = self.opt
        sched = None if self.sched is None else {k: v for k, v in vars(self.sched).items() if isinstance(v, (int, float))}
        return {"params": [p.detach().clone() for p in o.params], "step": o.step_count, "lr": o.lr, "sched": sched,
                "moments": None if o.master is None else (o.master.clone(), o.m.clone(), o.v.clone())}

    @torch.no_grad()
    def _restore(self, s):
        o = self.opt
        for p, q in zip(o.params, s["params"]):       # parameters are views of the flat buffer by now: this restores it
            p.copy_(q)
        if s["moments"] is not None:
            for dst, src in zip((o.master, o.m, o.v), s["moments"]):
                dst.copy_(src)
        else:                                         # the optimizer was built by the warm-up: back to a fresh one
            o.master.copy_(o.flat_param.float())
            o.m.zero_()
            o.v.zero_()
        o.step_count = s["step"]
        o.dev_state[1] = s["step"]
        if self.sched is not None:
            for k, v in s["sched"].items():
                setattr(self.sched, k, v)
        o.lr = s["lr"]

    def _after(self, replayed=True):
        if replayed:
            self.opt.step_count += 1                  # host mirror of the device counter (checkpoints read it)
        if self.sched is not None:
            self.sched.step()                         # sets optimizer.lr -> one tiny device write, outside the graph

    def __call__(self):
        if self.graph is None:
            raise RuntimeError("GraphedTrainStep was closed")
        self.graph.replay()
        if self.graph_opt is not None:
            self.reducer.begin_step()
            self.reducer.finish()                     # SUM over the ranks; 1 / world is folded into the optimizer kernel
            self.graph_opt.replay()
        self._after()
        return self.loss

    def close(self):
        """Drop the graph and hand the step state back to the host (FusedAdamW.release_device_state): the optimizer, the
        model and anything else in the process then behave as before the graphed step existed."""
        self.graph = self.graph_opt = None
        self.opt.release_device_state()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def train_step(model, images, soft_targets, optimizer, scheduler=None, reducer=None):
    """One optimisation step; returns the (device, un-synchronised) loss."""
    if hasattr(optimizer, "begin_step"):
        optimizer.begin_step()           # device-resident step state (after a GraphedTrainStep): counters advance here
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
