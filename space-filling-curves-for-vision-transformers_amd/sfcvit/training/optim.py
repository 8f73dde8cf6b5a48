"""Fused clip_grad_norm_ + AdamW on flat buffers (src/training/train.py:165-166,
main.py:288-289 of the reference: clip to 1.0, then AdamW(lr 3e-4, wd 5e-5)).

The reference runs ~150 per-tensor norm kernels with a host sync
(`clip_grad_norm_(foreach=False)`) and a foreach AdamW.  Here every parameter that
receives a gradient is a view into one bf16 flat buffer, gradients accumulate into
a second flat buffer (`p.grad` are views of it), and a step is two kernels: one sum
of squares, one fused clip + AdamW with fp32 master weights and moments.  Nothing
synchronises with the host.
"""
import torch

from .. import ops


import os as _os
_WT_CACHE = _os.environ.get("SFCVIT_WT_CACHE", "1") != "0"


class FlatGradBuffer:
    """Parameters and gradients of a model as views into two flat buffers (device- and
    dtype-agnostic torch plumbing; FusedAdamW adds the HIP step on top).

    The layout is decided after the first backward: parameters whose .grad is still None then
    never get one in this model (MixerBlock.token_mix*, src/models/vit.py:269-272) and stay
    outside the buffers, exactly as torch.optim skips them."""

    def __init__(self, params, grad_scale=1.0):
        self.params = [p for p in params if p.requires_grad]
        self.grad_scale = grad_scale            # 1/world_size when grads arrive SUM-reduced
        self.flat_param = self.flat_grad = None
        self.active = None                      # parameters that receive gradients, flat order
        self.offsets = None
        self.epoch = 0                          # bumped by zero_grad: one in-place gradient write per parameter and step
        self._wt = None                         # transposed(): (flat buffer of W^T, tile table, {id(p): view}, graph task id)

    def _check(self, p):
        pass

    def _build(self, active=None):
        """`active` = the parameters that receive gradients; by default those that have one now (after the first
        backward), which leaves out parameters the forward never touches (mlp_mixer.token_mix*, vit.py:269-272)."""
        self.active = [p for p in self.params if p.grad is not None] if active is None else list(active)
        if not self.active:
            raise RuntimeError("step() before any backward")
        dev, dtype = self.active[0].device, self.active[0].dtype
        for p in self.active:
            self._check(p)
            if p.dtype != dtype:
                raise TypeError("all parameters must share one dtype")
        self._wt = None                              # (a rebuilt layout invalidates the transposed-weight table)
        self.offsets, off = [], 0
        for p in self.active:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8          # keep every view 16-byte aligned
        self.flat_param = torch.zeros(off, device=dev, dtype=dtype)
        self.flat_grad = torch.zeros(off, device=dev, dtype=dtype)
        for p, o in zip(self.active, self.offsets):
            k = p.numel()
            self.flat_param[o:o + k].copy_(p.detach().reshape(-1))
            if p.grad is not None:
                self.flat_grad[o:o + k].copy_(p.grad.reshape(-1))
            p.data = self.flat_param[o:o + k].view(p.shape)
            p.grad = self.flat_grad[o:o + k].view(p.shape)
            p._sfcvit_slot = (self, o)           # functional._slot: kernels write this parameter's gradient in place
        self._built()

    def _built(self):
        pass

    def transposed(self, p):
        """W^T (contiguous [in, out]) of weight `p` for the dX GEMM of the running backward pass, or None.

        The dX GEMMs want both operands k-contiguous, i.e. a transposed copy of the weight: 51 launches of 5 us per
        ViT-B step when every GEMM makes its own (0.26 ms).  Here the first request inside a backward pass transposes EVERY
        2-D weight of the flat buffer in one launch (sfcvit_transpose_batched, 218 MB each way at ViT-B); the copies are
        valid for that backward pass only -- keyed by autograd's graph-task id, inside which the weights cannot change --
        so there is no invalidation rule to get wrong.  Outside a backward pass, for parameters outside the flat buffer,
        or with SFCVIT_WT_CACHE=0: None (the caller transposes by itself)."""
        task = ops.graph_task_id()
        if task < 0 or self.flat_param is None or not self.flat_param.is_cuda or not _WT_CACHE:
            return None
        if self._wt is None:
            import numpy as np
            mats, off = [], 0
            for q, o in zip(self.active, self.offsets):
                if q.dim() == 2 and q.shape[0] % 8 == 0 and q.shape[1] % 8 == 0 and q.shape[0] >= 64 and q.shape[1] >= 64:
                    mats.append((q, o, off))
                    off += q.numel()
            if not mats:
                self._wt = (None, None, {}, -1)
                return None
            rows = []
            for q, o, d in mats:
                R, C = q.shape
                rows += [(o, d, R, C, r0, c0) for r0 in range(0, R, 64) for c0 in range(0, C, 64)]
            table = np.array(rows, dtype=np.dtype([("src_off", "<i8"), ("dst_off", "<i8"), ("R", "<i4"), ("C", "<i4"),
                                                   ("r0", "<i4"), ("c0", "<i4")]))          # sfcvit_transpose_tile
            flat = torch.empty(off, device=self.flat_param.device, dtype=self.flat_param.dtype)
            tiles = torch.from_numpy(table.view(np.uint8).copy()).to(self.flat_param.device)
            views = {id(q): flat[d:d + q.numel()].view(q.shape[1], q.shape[0]) for q, o, d in mats}
            self._wt = [flat, (tiles, len(rows)), views, -1]
        flat, tiles, views, have = self._wt
        v = views.get(id(p))
        if v is None:
            return None
        if have != task:
            ops.transpose_batched(self.flat_param, flat, tiles[0], tiles[1])
            self._wt[3] = task
        return v

    def grad_views(self):
        """[(offset, numel, parameter)] in flat order (used by the data-parallel reducer)."""
        return [(o, p.numel(), p) for p, o in zip(self.active, self.offsets)]

    def zero_grad(self, set_to_none=True):
        """With set_to_none (default, as torch.optim) every p.grad is dropped: backward then writes gradients straight into
        the flat buffer and autograd adopts those views (functional._slot); gradients that arrive as separate tensors are
        moved in by `adopt` before they are used, and the slot of a parameter that received none is zeroed there -- every
        slot is overwritten in full each step, so the buffer itself (218 MB at ViT-B: a 40-us fill per step) is not
        cleared.  set_to_none=False keeps the views as p.grad and zeroes the buffer."""
        if self.flat_grad is None:
            for p in self.params:
                p.grad = None
            return
        self.epoch += 1
        if set_to_none:
            for p in self.active:
                p.grad = None
        else:
            self.flat_grad.zero_()

    def adopt(self, p, o=None):
        """Make p.grad the view of the flat buffer again (copying a gradient that was produced elsewhere)."""
        if o is None:
            o = p._sfcvit_slot[1]
        view = self.flat_grad[o:o + p.numel()].view(p.shape)
        g = p.grad
        if g is None:
            view.zero_()                         # no gradient this step
            p.grad = view
        elif g.data_ptr() != view.data_ptr():
            view.copy_(g)
            p.grad = view

    def adopt_all(self):
        for p, o in zip(self.active, self.offsets):
            self.adopt(p, o)


class FusedAdamW(FlatGradBuffer):
    """Not a torch.optim.Optimizer subclass on purpose: state is three flat fp32
    buffers, not per-tensor dicts.  API: zero_grad(), step(), lr attribute,
    state_dict() / load_state_dict()."""

    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-5, max_grad_norm=1.0,
                 grad_scale=1.0):
        super().__init__(params, grad_scale)
        # one group, shaped like torch.optim's so that schedulers written against `param_groups[0]['lr']` work
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self.master = self.m = self.v = self._sumsq = None
        self.dev_state = None            # use_device_state(): int32[8] device tensor (include/sfcvit.h, sfcvit_step_advance)
        self._seed_base = 0
        self._advanced_eagerly = False   # begin_step() ran outside a stream capture: step() then counts on the host too

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        self.param_groups[0]["lr"] = value
        if self.dev_state is not None:
            self.dev_state.view(torch.float32)[2].fill_(float(value))      # stream-ordered, no host sync

    def use_device_state(self, device=None, seed_base=None):
        """Move the per-step scalars to the device: Adam's step count and bias corrections, the learning rate, and an
        offset every dropout site adds to its seed.  `advance()` (one tiny kernel) then replaces the host-side
        `step_count += 1` / fresh seeds, so a captured training step is replayable (GraphedTrainStep)."""
        if self.dev_state is None:
            if device is None:
                device = self.params[0].device
            self.dev_state = torch.zeros(8, device=device, dtype=torch.int32)
            self.dev_state[1] = self.step_count
            self._seed_base = int(torch.randint(0, 0x7FFFFFFF, (), dtype=torch.int64)) if seed_base is None else int(seed_base)
            self.lr = self.lr                                               # writes state[2]
        ops.STEP_STATE = self.dev_state
        return self.dev_state

    def release_device_state(self):
        """Back to host-side step state (the default): later models / optimizers of this process draw their dropout seeds
        from torch's generator again instead of this optimizer's device offset (ops.STEP_STATE is process-global).  A
        hipGraph captured with the device state must not be replayed afterwards; GraphedTrainStep.close() calls this."""
        if self.dev_state is not None:
            if ops.STEP_STATE is self.dev_state:
                ops.STEP_STATE = None
            self.dev_state = None            # step_count has mirrored the device counter all along

    def advance(self):
        """Device-state mode: begin a training step (step += 1, new dropout seed offset, bias corrections)."""
        ops.step_advance(self.dev_state, self.betas[0], self.betas[1], self._seed_base)

    def begin_step(self):
        """First call of EVERY training step (train_step, the epoch loops and GraphedTrainStep make it).  Host-state
        mode: nothing to do.  Device-state mode: advance the device counters; when that happens eagerly (not inside a
        stream capture -- e.g. an odd-sized last batch that cannot replay the graph) step() also keeps the host mirror
        of the step count in line, so checkpoints and later replays agree with the device."""
        if self.dev_state is not None:
            self.advance()
            self._advanced_eagerly = not torch.cuda.is_current_stream_capturing()

    def _check(self, p):
        if p.dtype != torch.bfloat16 or not p.is_cuda:
            raise TypeError("FusedAdamW needs bf16 CUDA parameters (model.to('cuda', torch.bfloat16))")

    def _built(self):
        self.master = self.flat_param.float()
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        self._sumsq = torch.zeros(1, device=self.master.device, dtype=torch.float32)

    @torch.no_grad()
    def step(self):
        if self.flat_grad is None:
            self._build()
        else:
            self.adopt_all()
        if self.dev_state is None or self._advanced_eagerly:
            self.step_count += 1         # device-state mode, captured: the replay's advance() counts and GraphedTrainStep mirrors it
            self._advanced_eagerly = False
        sumsq = None
        if self.max_grad_norm is not None:
            self._sumsq.zero_()
            ops.sumsq_accum(self.flat_grad, self._sumsq)
            sumsq = self._sumsq
        ops.adamw_step(self.flat_param, self.master, self.flat_grad, self.m, self.v, sumsq,
                       lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                       weight_decay=self.weight_decay, max_norm=self.max_grad_norm or 0.0,
                       step=max(1, self.step_count), grad_scale=self.grad_scale, dev_state=self.dev_state)

    def grad_norm(self):
        """Total gradient norm of the last step (device tensor; no sync)."""
        return self._sumsq.sqrt() * self.grad_scale

    def state_dict(self):
        index = {id(p): i for i, p in enumerate(self.params)}
        return {"step": self.step_count, "lr": self.lr, "master": self.master, "m": self.m, "v": self.v,
                "active": [index[id(p)] for p in self.active]}

    def load_torch_state_dict(self, sd):
        """The state_dict of a torch.optim.AdamW over the same model.parameters() (what the reference's main.py:345-354
        checkpoints hold): per-parameter `step`, `exp_avg`, `exp_avg_sq` are mapped into the flat m / v buffers and the
        step count, lr from its param group.  Parameters without a state entry there (never stepped: the token-mix branch
        the forward does not use, vit.py:269-272) stay outside the flat buffers, as here.  The fp32 master weights are
        taken from the model's (already loaded) parameters: torch.optim.AdamW keeps none."""
        state, groups = sd["state"], sd["param_groups"]
        order = [i for g in groups for i in g["params"]]
        if len(order) != len(self.params):
            raise ValueError(f"optimizer state is for {len(order)} parameters, this model has {len(self.params)}")
        pos = {i: n for n, i in enumerate(order)}            # torch's parameter id -> position in model.parameters()
        active = sorted(pos[i] for i in state)
        if self.flat_grad is None:
            self._build([self.params[n] for n in active])
        elif [id(p) for p in self.active] != [id(self.params[n]) for n in active]:
            raise ValueError("optimizer state covers a different set of parameters than the one already built")
        steps = set()
        for i, st in state.items():
            p = self.params[pos[i]]
            o = p._sfcvit_slot[1]
            if st["exp_avg"].numel() != p.numel():
                raise ValueError(f"optimizer state of parameter {pos[i]} has {st['exp_avg'].numel()} elements, expected {p.numel()}")
            self.m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1).float())
            self.v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1).float())
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"parameters were stepped a different number of times ({sorted(steps)}): one flat step count cannot hold that")
        self.step_count = steps.pop() if steps else 0
        if self.dev_state is not None:
            self.dev_state[1] = self.step_count
        self.lr = float(groups[0]["lr"])
        self.master.copy_(self.flat_param.float())

    def load_state_dict(self, sd):
        if "param_groups" in sd and "state" in sd:           # a torch.optim optimizer's state (reference checkpoint)
            return self.load_torch_state_dict(sd)
        if self.flat_grad is None:
            self._build([self.params[i] for i in sd["active"]])
        if self.master.numel() != sd["master"].numel():
            raise ValueError("optimizer state does not fit these parameters")
        self.step_count, self.lr = sd["step"], sd["lr"]
        if self.dev_state is not None:
            self.dev_state[1] = self.step_count          # device-state mode: the kernel-side step counter follows
        self.master.copy_(sd["master"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.flat_param.copy_(self.master)


class WarmupCosine:
    """Linear warm-up then cosine decay, per step (transformers.get_cosine_schedule_with_warmup
    as used at main.py:310-314; half a cosine period, floor 0)."""

    def __init__(self, optimizer, warmup_steps, total_steps, base_lr=None):
        import math
        self.opt, self.warmup, self.total = optimizer, warmup_steps, total_steps
        self.base = optimizer.lr if base_lr is None else base_lr
        self.n = 0
        self._cos = math.cos
        self._pi = math.pi
        self.opt.lr = self.lr_at(0)

    def lr_at(self, n):
        if n < self.warmup:
            return self.base * n / max(1, self.warmup)
        prog = (n - self.warmup) / max(1, self.total - self.warmup)
        return self.base * max(0.0, 0.5 * (1.0 + self._cos(self._pi * prog)))

    def step(self):
        self.n += 1
        self.opt.lr = self.lr_at(self.n)


class WarmupCosineScheduler:
    """src/training/scheduler.py:4-51: linear warm-up to base_lr, cosine decay to min_lr; step() sets the rate for
    the step about to run, returns it, then advances.  Works on torch optimizers and on FusedAdamW."""

    def __init__(self, optimizer, warmup_steps, total_steps, min_lr=1e-6, base_lr=None):
        import math
        self._math = math
        self.optimizer = optimizer
        self.warmup_steps = warmup_steps
        self.total_steps = total_steps
        self.min_lr = min_lr
        self.base_lr = optimizer.param_groups[0]["lr"] if base_lr is None else base_lr
        self.current_step = 0

    def step(self):
        if self.current_step < self.warmup_steps:
            lr = self.base_lr * (self.current_step / max(1, self.warmup_steps))
        else:
            progress = (self.current_step - self.warmup_steps) / max(1, self.total_steps - self.warmup_steps)
            lr = self.min_lr + 0.5 * (self.base_lr - self.min_lr) * (1 + self._math.cos(self._math.pi * min(1.0, progress)))
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        self.current_step += 1
        return lr
