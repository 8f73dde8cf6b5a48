"""Data-parallel gradient reduction: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) across the xGMI mesh.

The reference has no distributed code; the path shards by images, so
the only exchange per step is the gradient sum.  Gradients already live in ONE flat
bf16 buffer (FusedAdamW), so a bucket is a contiguous slice of it: no copies in or
out.  Buckets are cut in flat order and all-reduced asynchronously as soon as every
parameter inside has accumulated its gradient (post-accumulate-grad hooks), which
overlaps the collectives with the rest of backward.  Reduction is SUM; the 1/world
factor is folded into the optimizer kernel (grad_scale), so clip-by-global-norm sees
the averaged gradient without an extra pass or a second collective.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): ViT-B has 218 MB of bf16
gradients per step, i.e. ~0.4-2.5 ms on the wire against >= 20 ms of backward, so a
few large buckets (default 32 MB: the last one, which cannot overlap with anything, costs ~0.2 ms) keep per-collective latency negligible and every
link busy.
"""
import os
import sys
import time

import torch
import torch.distributed as dist

_DEBUG = os.environ.get("SFCVIT_DDP_DEBUG", "0") == "1"      # one stderr line per collective launched / waited for


def _dbg(msg):
    if _DEBUG:
        print(f"[ddp r{dist.get_rank() if dist.is_initialized() else 0} {time.time() % 1000:8.3f}] {msg}", file=sys.stderr, flush=True)


class GradReducer:
    def __init__(self, optimizer, bucket_bytes=32 << 20, group=None):
        self.opt = optimizer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.buckets = None          # [(start, end)] element ranges of flat_grad
        self._pending = None         # per-bucket count of parameters still to arrive
        self._handles = []
        self._hooks = []
        self._param_bucket = {}
        self._wait_events = None     # reset_stats(): (before, after) event pairs around finish()'s waits
        optimizer.grad_scale = 1.0 / self.world

    # -- bucket plan over the flat gradient buffer ------------------------------------------
    @staticmethod
    def plan(sizes, bucket_elems):
        """Cut consecutive tensors (element counts `sizes`, each padded to a multiple of 8 as in
        the flat layout) into buckets of at most ~bucket_elems; returns [(first, last_exclusive)]
        tensor index ranges."""
        out, start, acc = [], 0, 0
        for i, n in enumerate(sizes):
            n = (n + 7) // 8 * 8
            if acc and acc + n > bucket_elems:
                out.append((start, i))
                start, acc = i, 0
            acc += n
        if acc:
            out.append((start, len(sizes)))
        return out

    def _install(self):
        views = self.opt.grad_views()
        elems = max(1, self.bucket_bytes // self.opt.flat_grad.element_size())
        ranges = self.plan([n for _, n, _ in views], elems)
        self.buckets, self._members = [], []
        total = self.opt.flat_grad.numel()
        for first, last in ranges:
            s = views[first][0]
            e = views[last][0] if last < len(views) else total
            self.buckets.append((s, e))
            self._members.append(last - first)
            for _, _, p in views[first:last]:
                self._param_bucket[id(p)] = len(self.buckets) - 1
        for _, _, p in views:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        if self.world > 1:
            # replicas must start identical: rank 0's parameters win (same-seed construction already gives that; a
            # model built or loaded differently on some rank would otherwise drift silently)
            _dbg("broadcast parameters")
            dist.broadcast(self.opt.flat_param, src=0, group=self.group)
            _dbg("broadcast done")
            master = getattr(self.opt, "master", None)
            if master is not None:
                master.copy_(self.opt.flat_param)

    def begin_step(self):
        self._handles = []
        if self.buckets is not None:
            self._pending = list(self._members)

    def _on_grad(self, p):
        if self._pending is None:
            return
        self.opt.adopt(p)                        # the bucket is a slice of the flat buffer: the gradient must be in it
        b = self._param_bucket[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self.world == 1:
            return
        s, e = self.buckets[b]
        _dbg(f"launch bucket {b} [{s}:{e}] (#{len(self._handles)} of this step)")
        self._handles.append(dist.all_reduce(self.opt.flat_grad[s:e], op=dist.ReduceOp.SUM,
                                             group=self.group, async_op=True))

    def finish(self):
        """Block the compute stream on every outstanding collective.  On the first step (flat
        layout not built yet) the gradients are still separate tensors: build, reduce once,
        install the hooks."""
        if self.buckets is None:
            if self.opt.flat_grad is None:
                self.opt._build()
            self._install()
            if self.world > 1:
                for b in range(len(self.buckets)):
                    self._launch(b)
        elif self._pending is not None and any(self._pending):
            # a parameter did not fire (e.g. frozen this step): reduce its bucket anyway
            for b, left in enumerate(self._pending):
                if left:
                    self._launch(b)
        timed = self._wait_events is not None and self.world > 1 and self._handles
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for i, h in enumerate(self._handles):
            _dbg(f"wait #{i}")
            h.wait()
        _dbg("step reduced")
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))
        self._handles = []
        self._pending = None

    # -- measurement (bench.py --gpus N) ------------------------------------------------------
    def reset_stats(self):
        """Start recording how long the compute stream is held at finish() by collectives that backward did not hide."""
        self._wait_events = []

    def stats(self, steps, reps=5):
        """Per-step figures for the bench line: bytes reduced, the time the compute stream spent blocked on the
        collectives (= the exposed, non-overlapped part), and -- measured afterwards with nothing else running -- the
        time and bus bandwidth of the same bucket sequence alone (bus GB/s = 2 (w - 1) / w * bytes / time, the
        ring-equivalent figure per GPU)."""
        torch.cuda.synchronize()
        exposed = sum(a.elapsed_time(b) for a, b in (self._wait_events or [])) / max(1, steps)
        self._wait_events = None
        out = {"buckets": len(self.buckets or []), "bucket_bytes": self.bucket_bytes,
               "bytes_per_step": int(self.opt.flat_grad.numel() * self.opt.flat_grad.element_size()) if self.buckets else 0,
               "exposed_ms_per_step": round(exposed, 4)}
        if self.world > 1 and self.buckets:
            scratch = torch.zeros_like(self.opt.flat_grad)
            def run():
                hs = [dist.all_reduce(scratch[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for s, e in self.buckets]
                for h in hs:
                    h.wait()
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out["allreduce_alone_ms"] = round(ms, 4)
            out["bus_gbs_alone"] = round(2 * (self.world - 1) / self.world * out["bytes_per_step"] / (ms * 1e-3) / 1e9, 1)
            out["overlapped_frac"] = round(max(0.0, 1.0 - exposed / ms), 4) if ms > 0 else None
        return out

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
