#!/usr/bin/env python3
"""Repository-free repro of the round-2 four-rank stall (gpurun_out/g4b.err) -- REWRITTEN: the script that
produced g4b.err (`tools/_g4b.py`) was not kept; this one has the same structure (an MLP whose parameters all-reduce from
post-accumulate-grad hooks over gloo, CUDA tensors, N ranks sharing card 0) and names the call the old line 25 was: the
`h.wait()` loop after backward (marked WAIT below) -- the faulthandler dump of g4b.err shows the main thread of all four
ranks in that statement after 40 s, and sfcvit's own trace (r2_4rank.err) shows the same for GradReducer.finish():
every rank launched the same eight buckets in the same order from its hooks and none returned from `wait #0`.

Unlike the original, every wait here is BOUNDED (10 s) and the process leaves after 60 s whatever happens, so running it
cannot hold a GPU lease.  Do not run it with more than 2 ranks on a shared pool without a reason: the stall is
reproducible evidence already in hand, and sfcvit no longer takes this path (GradReducer stages gloo through host memory
and refuses > 2 gloo ranks per card).

    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 tools/ddp_gloo_shared_gpu_repro.py
"""
import datetime
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(60, exit=True)
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=10))
rank = dist.get_rank()
torch.cuda.set_device(0)                                     # every rank on card 0
model = torch.nn.Sequential(*[torch.nn.Linear(4096, 4096) for _ in range(8)]).cuda()
handles = []
for p in model.parameters():
    p.register_post_accumulate_grad_hook(lambda q: handles.append(dist.all_reduce(q.grad, async_op=True)))
x = torch.randn(8192, 4096, device="cuda")
for step in range(3):
    t0 = time.time()
    handles.clear()
    model.zero_grad()
    model(x).square().mean().backward()                      # hooks launch the all-reduces while backward is still being enqueued
    for i, h in enumerate(handles):
        try:
            h.wait(datetime.timedelta(seconds=10))           # WAIT: the statement the four ranks of g4b.err were parked in
        except RuntimeError as e:
            print(f"rank {rank} step {step}: wait #{i} of {len(handles)} timed out after {time.time() - t0:.1f} s: {e}", file=sys.stderr, flush=True)
            os._exit(3)
    torch.cuda.synchronize()
    print(f"rank {rank} step {step}: {time.time() - t0:.2f} s", flush=True)
dist.destroy_process_group()
