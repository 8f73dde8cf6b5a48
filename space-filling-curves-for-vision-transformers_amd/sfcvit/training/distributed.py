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
import datetime
import faulthandler
import os
import sys
import threading
import time

import torch
import torch.distributed as dist

_DEBUG = os.environ.get("SFCVIT_DDP_DEBUG", "0") == "1"      # one stderr line per collective launched / waited for
# Every wait on a collective is bounded: a rank that does not get its bucket within this many seconds reports which
# bucket of which step it was waiting for and leaves with a non-zero exit code (a stuck collective otherwise holds the
# whole lease: VERDICT r2 #5).  Also handed to init_process_group by bench.py / main.py (dist_timeout()).
# The bound is on a collective that makes NO progress: the RCCL watcher counts from the later of a collective's launch and the
# last completion of ANY collective of this reducer, so a slow peer that is still moving does not trip it; a peer that
# stays away from a step altogether (a long checkpoint write or an evaluation only one rank runs) does once it has been
# away for this long -- all ranks must reach every step within SFCVIT_DIST_TIMEOUT of each other.  600 s = NCCL's own default.
DIST_TIMEOUT_S = float(os.environ.get("SFCVIT_DIST_TIMEOUT", "600"))
# Gradients are summed across ranks in the flat buffer's dtype (bf16: 218 MB per ViT-B step on the wire).  A ring sums the
# world's addends hop by hop with one bf16 rounding per hop; tests/test_distributed_cpu.py measures that at world 8
# (relative L2 error 3.3e-3 of the exact sum, against 1.7e-3 for one final rounding -- far inside the 2.5 % gradient-norm
# envelope the single-GPU step is held to).  SFCVIT_DDP_FP32=1 (or reduce_dtype=torch.float32) sums fp32 copies of the
# buckets instead: twice the bytes, one rounding.
REDUCE_FP32 = os.environ.get("SFCVIT_DDP_FP32", "0") == "1"


def dist_timeout():
    return datetime.timedelta(seconds=DIST_TIMEOUT_S)


class CollectiveTimeout(RuntimeError):
    pass


def check_rehearsal_layout(backend, world, ranks_on_this_device):
    """gloo ranks that SHARE one GPU are a functional rehearsal of the data-parallel path on a one-GPU box, nothing
    more; more than two of them on a card are refused: the launch that tried four (gpurun_out/r2_4rank.err)
    sat in its first hook-launched all-reduce until the lease ended."""
    if backend == "gloo" and ranks_on_this_device > 2:
        raise RuntimeError(f"{ranks_on_this_device} gloo ranks on one GPU (world {world}): refused.  The gloo rehearsal is "
                           "supported for at most 2 ranks per device (DESIGN.md §6); use the CPU tests for more ranks, or one "
                           "GPU per rank with the default backend (nccl = RCCL).")


def _dbg(msg):
    if _DEBUG:
        print(f"[ddp r{dist.get_rank() if dist.is_initialized() else 0} {time.time() % 1000:8.3f}] {msg}", file=sys.stderr, flush=True)


class GradReducer:
    """overlap=True (default): buckets are all-reduced from post-accumulate-grad hooks while backward is still running.
    overlap=False: no hooks -- finish() launches every bucket itself.  That is the mode a GraphedTrainStep needs (forward
    + backward replay from one hipGraph, in which hooks cannot run; the collectives are issued between that graph and
    the optimizer's), and it costs the overlap with backward, not correctness: same buckets, same sums."""

    def __init__(self, optimizer, bucket_bytes=32 << 20, group=None, overlap=True, reduce_dtype=None):
        self.opt = optimizer
        self.reduce_dtype = reduce_dtype if reduce_dtype is not None else (torch.float32 if REDUCE_FP32 else None)
        self._last_progress = time.monotonic()
        self.group = group
        self.overlap = overlap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.buckets = None          # [(start, end)] element ranges of flat_grad
        self._pending = None         # per-bucket count of parameters still to arrive
        self._handles = []
        self._hooks = []
        self._param_bucket = {}
        self._wait_events = None     # reset_stats(): (before, after) event pairs around finish()'s waits
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.timeout_s = DIST_TIMEOUT_S
        self.step_no = 0
        # gloo has no device path worth the name: given CUDA tensors it stages them through host memory on threads and
        # streams of its own, and with hook-launched collectives racing the rest of backward for the card that is where
        # the four-rank rehearsal stalled.  Under gloo the reducer therefore stages the flat gradient
        # through ONE pinned host buffer itself, at finish(), bucket by bucket with blocking, bounded waits:
        # no overlap (it is a rehearsal), the same buckets, the same arithmetic, the same result on every rank.
        self._host_staged = None     # decided at _install (needs the flat buffer's device)
        self._host_buf = None
        self._watch = None           # RCCL: [(step, bucket, handle, t_launch)] watched by a daemon thread
        self._watch_lock = threading.Lock()
        optimizer.grad_scale = 1.0 / self.world

    # -- bucket plan over the flat gradient buffer ------------------------------------------
    @staticmethod
    def plan(sizes, bucket_elems, first_elems=None):
        """Cut consecutive tensors (element counts `sizes`, each padded to a multiple of 8 as in
        the flat layout) into buckets of at most ~bucket_elems; returns [(first, last_exclusive)]
        tensor index ranges.  first_elems: a smaller limit for the FIRST bucket -- the parameters the forward uses first
        get their gradients last, so this bucket's all-reduce starts when backward ends and nothing is left to hide it under:
        its wire time is the exposed part of the exchange, and a quarter-size bucket makes that a quarter."""
        out, start, acc = [], 0, 0
        for i, n in enumerate(sizes):
            n = (n + 7) // 8 * 8
            limit = first_elems if first_elems is not None and not out else bucket_elems
            if acc and acc + n > limit:
                out.append((start, i))
                start, acc = i, 0
            acc += n
        if acc:
            out.append((start, len(sizes)))
        return out

    def _install(self):
        views = self.opt.grad_views()
        elems = max(1, self.bucket_bytes // self.opt.flat_grad.element_size())
        ranges = self.plan([n for _, n, _ in views], elems, first_elems=max(1, elems // 4))
        self.buckets, self._members = [], []
        total = self.opt.flat_grad.numel()
        for first, last in ranges:
            s = views[first][0]
            e = views[last][0] if last < len(views) else total
            self.buckets.append((s, e))
            self._members.append(last - first)
            for _, _, p in views[first:last]:
                self._param_bucket[id(p)] = len(self.buckets) - 1
        if self.overlap:
            for _, _, p in views:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._host_staged = self.world > 1 and self.backend == "gloo" and self.opt.flat_grad.is_cuda
        self.sync_replicas()

    def sync_replicas(self):
        """Replicas must start identical: rank 0's parameters win (same-seed construction already gives that; a model
        built or loaded differently on some rank would otherwise drift silently).  With fp32 master weights it is the
        MASTER that is broadcast and the bf16 parameters follow from it: after `--resume` rank 0's master carries the
        sub-bf16-ulp part of the checkpointed weights, which a broadcast of the rounded bf16 parameters followed by
        master <- param would throw away on every rank.  Called at the first reduced step, and again by
        GraphedTrainStep after it has put every rank's own pre-warm-up state back."""
        if self.world > 1 and self.opt.flat_param is not None:
            master = getattr(self.opt, "master", None)
            src = master if master is not None else self.opt.flat_param
            _dbg("broadcast parameters" + (" (fp32 master)" if master is not None else ""))
            self._broadcast(src)
            _dbg("broadcast done")
            if master is not None:
                self.opt.flat_param.copy_(master)

    def _broadcast(self, t):
        if self.backend == "gloo" and t.is_cuda:          # host-staged, as the gradient buckets
            h = t.cpu()
            self._bounded(dist.broadcast(h, src=0, group=self.group, async_op=True), "parameter broadcast")
            t.copy_(h)
        else:
            dist.broadcast(t, src=0, group=self.group)

    def _bounded(self, work, what):
        """Host-blocking wait with the reducer's timeout; raises CollectiveTimeout naming `what`."""
        try:
            ok = work.wait(datetime.timedelta(seconds=self.timeout_s))
        except RuntimeError as e:                            # gloo raises on timeout
            raise CollectiveTimeout(f"rank {dist.get_rank()}: {what} did not complete within {self.timeout_s:.0f} s "
                                    f"(SFCVIT_DIST_TIMEOUT): {e}") from e
        if ok is False:
            raise CollectiveTimeout(f"rank {dist.get_rank()}: {what} did not complete within {self.timeout_s:.0f} s")

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
        if self._host_staged:                    # gloo rehearsal: reduced at finish(), through the host buffer
            self._handles.append((b, None))
            return
        _dbg(f"launch bucket {b} [{s}:{e}] (#{len(self._handles)} of this step)")
        if self.opt.flat_grad.is_cuda:
            from .. import ops
            ops.flush_deferred(end=False)        # reductions queued for the end of the backward pass end in this bucket
        g = self.opt.flat_grad[s:e]
        wide = None
        if self.reduce_dtype is not None and self.reduce_dtype != g.dtype:
            wide = g.to(self.reduce_dtype)       # summed in fp32, rounded once when finish() copies it back
            g = wide
        h = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append((b, h) if wide is None else (b, h, wide))
        if self.backend == "nccl":
            self._watch_add(b, h)

    # -- RCCL: waits are stream waits (the host runs ahead), so the bound is kept by a watcher thread ----------------
    def _watch_add(self, b, h):
        with self._watch_lock:
            if self._watch is None:
                self._watch = []
                threading.Thread(target=self._watcher, name="sfcvit-collective-watch", daemon=True).start()
            self._watch.append((self.step_no, b, h, time.monotonic()))

    def _watcher(self):
        while True:
            time.sleep(1.0)
            now = time.monotonic()
            with self._watch_lock:
                n = len(self._watch)
                self._watch = [w for w in self._watch if not w[2].is_completed()]
                if len(self._watch) < n:
                    self._last_progress = now        # something completed: the exchange is alive
                late = [w for w in self._watch if now - max(w[3], self._last_progress) > self.timeout_s]
            if late:
                step, b, _, t0 = late[0]
                s, e = self.buckets[b]
                print(f"[sfcvit] rank {dist.get_rank()}: all-reduce of gradient bucket {b} [{s}:{e}] launched in step {step} has "
                      f"not completed after {now - t0:.0f} s (SFCVIT_DIST_TIMEOUT = {self.timeout_s:.0f}); {len(late)} collective(s) "
                      "overdue.  Exiting with code 4.", file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                os._exit(4)

    def finish(self):
        """Block the compute stream on every outstanding collective.  On the first step (flat
        layout not built yet) the gradients are still separate tensors: build, reduce once,
        install the hooks."""
        if self.buckets is None:
            if self.opt.flat_grad is None:
                self.opt._build()
            else:
                self.opt.adopt_all()             # layout built earlier (load_state_dict): gradients produced outside the slots move in
            self._install()
            if self.world > 1:
                for b in range(len(self.buckets)):
                    self._launch(b)
        elif not self.overlap:
            self.opt.adopt_all()                 # gradients produced outside the slots move into the flat buffer first
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
        if self._host_staged and self._handles:
            self._reduce_through_host()
        else:
            for i, (b, h, *wide) in enumerate(self._handles):
                _dbg(f"wait #{i} (bucket {b})")
                if self.backend == "nccl":
                    h.wait()                     # stream wait only; bounded by the watcher thread
                else:
                    self._bounded(h, f"all-reduce of gradient bucket {b} {list(self.buckets[b])} (collective #{i} of step {self.step_no})")
                if wide:
                    s, e = self.buckets[b]
                    self.opt.flat_grad[s:e].copy_(wide[0])
        _dbg("step reduced")
        self.step_no += 1
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))
        self._handles = []
        self._pending = None

    def _reduce_through_host(self):
        g = self.opt.flat_grad
        if self._host_buf is None:
            self._host_buf = torch.empty(g.shape, dtype=g.dtype, pin_memory=True)
        order = [h[0] for h in self._handles]                 # the order the hooks completed the buckets in (same on every rank)
        self._host_buf.copy_(g)                               # D2H on the compute stream; blocks the host until it is there
        for i, b in enumerate(order):
            s, e = self.buckets[b]
            _dbg(f"host-staged all-reduce #{i}: bucket {b} [{s}:{e}]")
            piece = self._host_buf[s:e]
            if self.reduce_dtype is not None and self.reduce_dtype != piece.dtype:
                piece = piece.to(self.reduce_dtype)
            work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._bounded(work, f"all-reduce of gradient bucket {b} [{s}:{e}] (collective #{i} of step {self.step_no}, host-staged gloo)")
            if piece.dtype != self._host_buf.dtype:
                self._host_buf[s:e].copy_(piece)
        g.copy_(self._host_buf)

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
            scratch = torch.zeros_like(self._host_buf if self._host_staged else self.opt.flat_grad)
            def run():
                hs = [dist.all_reduce(scratch[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for s, e in self.buckets]
                for b, h in enumerate(hs):
                    if self.backend == "nccl":
                        h.wait()
                    else:
                        self._bounded(h, f"stand-alone all-reduce of bucket {b}")
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
