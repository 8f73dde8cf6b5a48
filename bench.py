#!/usr/bin/env python3
"""bench.py -- training throughput of the MI355X-native SFC-ViT hot path.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] / [3]): ViT-B/16 at 224x224, Hilbert pixel order
(HilbertEmbedding1D(224, 256, 3, 768) + VisionTransformer1D(depth 12, heads 12, mlp 3072,
1000 classes)), 256 synthetic images per GPU, training mode with the reference's dropout
(0.1 at the four encoder sites, 0.5 in the head), one full training step = zero_grad ->
forward -> soft-target CE -> backward -> (gradient all-reduce) -> clip 1.0 -> AdamW.
Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline`
(the dominant kernel, timed live with HIP events) and `cpu_baseline` (the oracle's
fp32 PyTorch-CPU restatement timed on the host cores, rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (img, patch_px, D, depth, heads, mlp, classes, per-GPU batch)
    "vit_b16_224_hilbert": (224, 256, 768, 12, 12, 3072, 1000, 256),
    "vit_tiny16_32_hilbert": (32, 256, 192, 12, 3, 768, 10, 256),
    "vit_l16_384_hilbert": (384, 256, 1024, 24, 16, 4096, 1000, 64),
}


def train_flops_per_image(img, patch, D, depth, heads, F, classes):
    """F_fwd = 2NKD + 8ND^2 + L(8ND^2 + 4N^2 D + 4NDF) + 2NDR + 4NRD + 4DC; train = 3F - 2NKD."""
    N, K, R = img * img // patch, 3 * patch, 64
    fwd = 2 * N * K * D + 8 * N * D * D + depth * (8 * N * D * D + 4 * N * N * D + 4 * N * D * F) \
        + 2 * N * D * R + 4 * N * R * D + 4 * D * classes
    return 3 * fwd - 2 * N * K * D


def build(workload, dropout):
    from sfcvit.models import VisionTransformer1D
    from sfcvit.tokenizers import HilbertEmbedding1D
    img, patch, D, depth, heads, mlp, classes, _ = WORKLOADS[workload]
    torch.manual_seed(42)                                   # main.py:151-152
    pe = HilbertEmbedding1D(img, patch, 3, D)
    model = VisionTransformer1D(pe, depth=depth, n_heads=heads, mlp_dim=mlp, num_classes=classes,
                                dropout_p=dropout, head_dropout_p=0.5 if dropout > 0 else 0.0)
    return model


def cpu_baseline(workload, batch, steps):
    """The oracle (fp32 PyTorch-CPU restatement of the reference path) on the host cores: full
    train step (fwd + soft-target CE + bwd + clip + AdamW), same synthetic input definition."""
    from oracle import vit_oracle
    img, patch, D, depth, heads, mlp, classes, _ = WORKLOADS[workload]
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = vit_oracle.OracleConfig("hilbert1d", img, patch, 3, D, depth, heads, mlp, classes, "1d")
    sd = vit_oracle.random_state(cfg, seed=42)
    leaves = [v.requires_grad_(True) for k, v in vit_oracle.trainable(sd).items()
              if not k.startswith(vit_oracle.UNUSED_PREFIXES)]
    opt = torch.optim.AdamW(leaves, lr=3e-4, weight_decay=5e-5)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(batch, 3, img, img, generator=g)
    y = torch.randint(0, classes, (batch,), generator=g)
    one = torch.nn.functional.one_hot(y, classes).float()
    tgt = 0.7 * one + 0.3 * one.roll(1, 0)
    vit_oracle.train_step(x, tgt, sd, cfg, opt)             # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        vit_oracle.train_step(x, tgt, sd, cfg, opt)
    dt = (time.perf_counter() - t0) / steps
    return {"value": batch / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} train steps of batch {batch} (fp32, torch {torch.__version__} CPU, "
                      f"{cores} threads) after 1 warm-up, same model and synthetic inputs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="vit_b16_224_hilbert", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--dropout", type=float, default=0.1,
                    help="encoder dropout (reference default 0.1; the head then uses the reference's 0.5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="HIP-event pairs around every op (default: only the GEMM families, the dominant kernels; "
                         "700 event pairs per step cost ~2.5 %% of throughput)")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # Rehearsal knobs (single-GPU box): SFCVIT_DIST_BACKEND=gloo + SFCVIT_FORCE_DEVICE=0 run N ranks of the
    # real data-parallel code path on one card; the driver's N-GPU runs use neither.
    backend = os.environ.get("SFCVIT_DIST_BACKEND", "nccl")
    local = int(os.environ.get("SFCVIT_FORCE_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from sfcvit import ops
    from sfcvit.training import FusedAdamW, GradReducer, mixup_soft_targets, train_step

    img, patch, D, depth, heads, mlp, classes, batch = WORKLOADS[args.workload]
    batch = args.batch or batch
    model = build(args.workload, args.dropout).to(dev, dtype=torch.bfloat16)
    model.train() if args.dropout > 0 else model.eval()    # eval() only switches dropout off; grads flow
    opt = FusedAdamW(model.parameters(), lr=3e-4, weight_decay=5e-5, max_grad_norm=1.0)
    reducer = GradReducer(opt) if world > 1 else None

    g = torch.Generator(device=dev).manual_seed(42 + rank)   # per-rank data, identical init
    images = torch.randn(batch, 3, img, img, device=dev, generator=g)
    labels = torch.randint(0, classes, (batch,), device=dev, generator=g)
    targets = mixup_soft_targets(labels, classes, lam=0.7)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = train_step(model, images, targets, opt, reducer=reducer)
    if rank == 0 and not args.no_kernel_timing:
        ops.TIMER = ops.KernelTimer(None if args.time_all_kernels else "gemm ")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(model, images, targets, opt, reducer=reducer)
    sync()
    dt = time.perf_counter() - t0
    kern = ops.TIMER.summary() if ops.TIMER is not None else {}
    ops.TIMER = None
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    loss_v = float(loss)
    if not (loss_v == loss_v):
        raise SystemExit("loss is NaN")

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * batch * args.steps / dt
        flops_img = train_flops_per_image(img, patch, D, depth, heads, mlp, classes)
        out = {
            "metric": "images/sec ViT-B/16 224px Hilbert-order training" if args.workload == "vit_b16_224_hilbert"
                      else f"images/sec {args.workload} training",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {img}x{img} Hilbert pixel order, 16x16-pixel tokens, "
                                   f"batch {batch}/GPU, full train step (fwd+CE+bwd+clip+AdamW)",
                       "global_batch": world * batch, "parallelism": f"dp{world}", "dropout": args.dropout},
            "train_gflop_per_image": round(flops_img / 1e9, 3),
            "step_tflops_per_gpu": round(value / world * flops_img / 1e12, 2),
            "step_mfma_frac": round(value / world * flops_img / 1e12 / PEAK_BF16_TFLOPS, 4),
            "final_loss": round(loss_v, 4),
        }
        if kern:
            gemm_keys = [k for k in kern if k.startswith("gemm ")]
            dom = max(kern, key=lambda k: kern[k]["ms_total"])
            r = kern[dom]
            ach = r["work_total"] / (r["ms_total"] * 1e-3) / 1e12
            # GEMM timing keys are "gemm <kernel symbol>" (sfcvit_last_gemm_kernel): the name rocprofv3 shows for the
            # same launches in profiles/ (there prefixed with the namespace).
            symbol = dom[5:] if dom.startswith("gemm ") else dom
            traffic = None                          # PMC counters cannot be read live: offline passes, see profiles/traffic.json
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    traffic = json.load(f)["kernels"].get(symbol, {}).get("traffic_bytes")
            except (OSError, ValueError, KeyError):
                pass
            out["roofline"] = {"kernel": dom, "rocprof_symbol": symbol, "bound": "mfma", "achieved": round(ach, 2),
                               "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                               "traffic": traffic, "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per MI355X_MICROARCH.md)" if traffic else None,
                               "launches": r["launches"], "avg_launch_ms": round(r["ms_avg"], 4),
                               "flops_per_launch": r["work_total"] / r["launches"]}
            out["kernels"] = {k: {"launches": v["launches"], "ms_per_step": round(v["ms_total"] / args.steps, 3),
                                  "tflops": round(v["work_total"] / (v["ms_total"] * 1e-3) / 1e12, 1)}
                              for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms_total"])}
            out["gemm_ms_per_step"] = round(sum(kern[k]["ms_total"] for k in gemm_keys) / args.steps, 3)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_batch, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
