#!/usr/bin/env python3
"""bench.py -- training throughput of the MI355X-native SFC-ViT hot path.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --cpu-only                      # BASELINE config 1: ViT-Tiny raster, batch 32, host cores only

Default workload (BASELINE.json configs[2] / [3]): ViT-B/16 at 224x224, Hilbert pixel order
(HilbertEmbedding1D(224, 256, 3, 768) + VisionTransformer1D(depth 12, heads 12, mlp 3072,
1000 classes)), 256 synthetic images per GPU, training mode with the reference's dropout
(0.1 at the four encoder sites, 0.5 in the head), one full training step = zero_grad ->
forward -> soft-target CE -> backward -> (gradient all-reduce) -> clip 1.0 -> AdamW.
Prints ONE JSON line on rank 0 (contract in the task statement).

What the line's roofline numbers are, so that they can be recomputed from profiles/ alone:
  * `roofline`      the dominant kernel FAMILY of the step: all instantiations of the persistent 8-phase GEMM
                    `gemm8p_kernel<NI, MASK, P2>` (every forward and dX GEMM).  achieved = sum of 2MNK over the family's
                    launches in the timed region / sum of their HIP-event durations (events on the launching stream);
                    peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).  `traffic` = launch-weighted mean of the
                    family's HBM-side bytes per launch from the PMC passes in profiles/traffic.json for THIS workload
                    (null when the workload has no PMC pass); `algorithmic_bytes` beside it.
  * `roofline_detail`  the same quantity for: every GEMM symbol, the worst GEMM, the weight-gradient kernel
                    (`gemm8p_km_kernel<P2>`), attention forward / backward (both bounds: FLOP/s of 2.5 PF and algorithmic
                    bytes/s of 8 TB/s -- at N = 196, hd = 64 attention is below the ridge, i.e. HBM-bound), the fused
                    gather + patch-embed kernels (both bounds) and the whole step.
  * `cpu_baseline`  the oracle's fp32 PyTorch-CPU restatement of the reference path timed on the host cores
                    (rank 0 at N = 1 only): BASELINE.md §3 -- 1 warm-up + 3 timed steps, all cores of the box's share.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# dmabuf IPC is the only kind this pool's host driver supports: without it RCCL (and any cross-process sharing of
# device memory) fails with "hipIpcGetMemHandle: invalid argument".  Must be in the environment before HIP initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (tokenizer, img, pixels per token, D, depth, heads, mlp, classes, per-GPU batch, CPU-baseline batch)
    "vit_b16_224_hilbert": ("hilbert", 224, 256, 768, 12, 12, 3072, 1000, 256, 32),      # BASELINE configs 3, 4
    "vit_tiny16_32_hilbert": ("hilbert", 32, 256, 192, 12, 3, 768, 10, 256, 256),         # config 2
    "vit_tiny16_32_raster": ("raster", 32, 256, 192, 12, 3, 768, 10, 32, 32),             # config 1 (CPU plumbing case)
    "vit_l16_384_hilbert": ("hilbert", 384, 256, 1024, 24, 16, 4096, 1000, 64, 8),        # config 5, three orders
    "vit_l16_384_z": ("z", 384, 256, 1024, 24, 16, 4096, 1000, 64, 8),
    "vit_l16_384_raster": ("raster", 384, 256, 1024, 24, 16, 4096, 1000, 64, 8),
}
ORDER_NAME = {"hilbert": "Hilbert", "z": "Z (Morton)", "raster": "raster"}


def train_flops_per_image(img, patch, D, depth, heads, F, classes):
    """F_fwd = 2NKD + 8ND^2 + L(8ND^2 + 4N^2 D + 4NDF) + 2NDR + 4NRD + 4DC; train = 3F - 2NKD."""
    N, K, R = img * img // patch, 3 * patch, 64
    fwd = 2 * N * K * D + 8 * N * D * D + depth * (8 * N * D * D + 4 * N * N * D + 4 * N * D * F) \
        + 2 * N * D * R + 4 * N * R * D + 4 * D * classes
    return 3 * fwd - 2 * N * K * D


def build(workload, dropout):
    from sfcvit.models import VisionTransformer1D
    from sfcvit.tokenizers import HilbertEmbedding1D, MortonEmbedding1D, RasterScan1DEmbedding
    tok, img, patch, D, depth, heads, mlp, classes, _, _ = WORKLOADS[workload]
    torch.manual_seed(42)                                   # main.py:151-152
    cls = {"hilbert": HilbertEmbedding1D, "z": MortonEmbedding1D, "raster": RasterScan1DEmbedding}[tok]
    pe = cls(img, patch, 3, D)
    model = VisionTransformer1D(pe, depth=depth, n_heads=heads, mlp_dim=mlp, num_classes=classes,
                                dropout_p=dropout, head_dropout_p=0.5 if dropout > 0 else 0.0)
    return model


def host_cores():
    """Cores this process may use: the affinity mask, capped by the cgroup CPU quota (the GPU box gives a one-GPU job a
    16-core share of a larger host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, quota // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, batch, steps, cores=None):
    """The oracle (fp32 PyTorch-CPU restatement of the reference path) on the host cores: full
    train step (fwd + soft-target CE + bwd + clip + AdamW), same synthetic input definition."""
    from oracle import vit_oracle
    tok, img, patch, D, depth, heads, mlp, classes, _, _ = WORKLOADS[workload]
    cores = cores or host_cores()
    torch.set_num_threads(cores)
    kind = {"hilbert": "hilbert1d", "z": "morton1d", "raster": "raster1d"}[tok]
    cfg = vit_oracle.OracleConfig(kind, img, patch, 3, D, depth, heads, mlp, classes, "1d")
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
    flops = train_flops_per_image(img, patch, D, depth, heads, mlp, classes)
    return {"value": batch / dt, "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "ms_per_step": dt * 1e3, "achieved_gflops": batch / dt * flops / 1e9,
            "sample": f"{steps} train steps of batch {batch} (fp32, torch {torch.__version__} CPU, "
                      f"{cores} threads) after 1 warm-up, same model and synthetic input definition"}


def load_traffic(workload):
    """profiles/traffic.json: {"workloads": {workload: {"kernels": {symbol: {"traffic_bytes": ...}}}}} -- per-launch
    HBM-side bytes from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/traffic_from_pmc.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            doc = json.load(f)
        wl = doc.get("workloads", {}).get(workload)
        return (wl or {}).get("kernels", {}), (wl or {}).get("build")
    except (OSError, ValueError):
        return {}, None


def frac_entry(rec, bound_flops=True, bytes_per_launch=None):
    """A roofline entry from a KernelTimer record (work_total = FLOPs)."""
    sec = rec["ms_total"] * 1e-3
    out = {"launches": rec["launches"], "avg_launch_ms": round(rec["ms_avg"], 4),
           "tflops": round(rec["work_total"] / sec / 1e12, 1),
           "mfma_frac": round(rec["work_total"] / sec / 1e12 / PEAK_BF16_TFLOPS, 4)}
    if bytes_per_launch is not None:
        gbs = bytes_per_launch * rec["launches"] / sec / 1e9
        out.update({"algorithmic_bytes": int(bytes_per_launch), "gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)})
        out["bound"] = "hbm" if out["hbm_frac"] >= out["mfma_frac"] else "mfma"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)      # ~2 s of GPU time at ViT-B: long enough for an external utilisation sampler to see it
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--dropout", type=float, default=0.1,
                    help="encoder dropout (reference default 0.1; the head then uses the reference's 0.5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="HIP-event pairs around every op (default: GEMMs, attention and patch embed; ~740 event pairs "
                         "per ViT-B step cost ~2.5 %% of throughput)")
    ap.add_argument("--time-every", type=int, default=4,
                    help="HIP-event timing of the kernels in every n-th timed step (1 = all: +3.9 %% step time at ViT-B; 4: +1 %%)")
    ap.add_argument("--cpu-only", action="store_true",
                    help="no GPU: time the CPU restatement only (default workload vit_tiny16_32_raster, BASELINE config 1)")
    ap.add_argument("--cpu-batch", type=int, default=0, help="CPU-baseline batch (default: BASELINE.md §3's per workload)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="replay the whole training step from one hipGraph")
    args = ap.parse_args()
    # Deadline: a run still going after this many seconds dumps every thread's Python stack to stderr and exits with
    # code 1 instead of holding the GPU lease (SFCVIT_BENCH_WATCHDOG; default for multi-rank runs: 900 s; 0 = off).
    watchdog = int(os.environ.get("SFCVIT_BENCH_WATCHDOG", "900" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "0"))
    if watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)

    if args.cpu_only:
        wl = args.workload or "vit_tiny16_32_raster"
        tok, img, patch, D, depth, heads, mlp, classes, _, cbatch = WORKLOADS[wl]
        base = cpu_baseline(wl, args.cpu_batch or cbatch, args.cpu_steps)
        print(json.dumps({"metric": f"images/sec {wl} training (CPU reference path)", "value": round(base["value"], 2),
                          "unit": "images/s", "n_gpus": 0, "steps": args.cpu_steps, "warmup": 1,
                          "ms_per_step": round(base["ms_per_step"], 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"{wl}: {img}x{img} {ORDER_NAME[tok]} pixel order, batch "
                                                 f"{args.cpu_batch or cbatch}, full train step on the host cores"},
                          "cpu_baseline": base}), flush=True)
        return

    args.workload = args.workload or "vit_b16_224_hilbert"
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
        from sfcvit.training.distributed import check_rehearsal_layout, dist_timeout
        # ranks that share this card: with SFCVIT_FORCE_DEVICE every rank of the node, otherwise one
        check_rehearsal_layout(backend, world, world if "SFCVIT_FORCE_DEVICE" in os.environ else 1)
        # every collective is bounded (SFCVIT_DIST_TIMEOUT, default 120 s): the backend's own watchdog gets the same limit
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=dist_timeout())
        else:
            dist.init_process_group(backend, timeout=dist_timeout())
        assert dist.get_world_size() == args.gpus, "process group size != --gpus"

    from sfcvit import ops
    from sfcvit.training import FusedAdamW, GradReducer, mixup_soft_targets, train_step

    tok, img, patch, D, depth, heads, mlp, classes, batch, cbatch = WORKLOADS[args.workload]
    batch = args.batch or batch
    N = img * img // patch
    model = build(args.workload, args.dropout).to(dev, dtype=torch.bfloat16)
    model.train() if args.dropout > 0 else model.eval()    # eval() only switches dropout off; grads flow
    opt = FusedAdamW(model.parameters(), lr=3e-4, weight_decay=5e-5, max_grad_norm=1.0)
    # with --graph the collectives run between two graphs (forward + backward | optimizer), not from autograd hooks
    reducer = GradReducer(opt, overlap=not args.graph) if world > 1 else None

    g = torch.Generator(device=dev).manual_seed(42 + rank)   # per-rank data, identical init
    images = torch.randn(batch, 3, img, img, device=dev, generator=g)
    labels = torch.randint(0, classes, (batch,), device=dev, generator=g)
    targets = mixup_soft_targets(labels, classes, lam=0.7)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = lambda: train_step(model, images, targets, opt, reducer=reducer)     # noqa: E731
    if args.graph:
        from sfcvit.training import GraphedTrainStep
        step = GraphedTrainStep(model, images, targets, opt, warmup=max(2, args.warmup), preserve_state=False, reducer=reducer)   # its warm-up steps ARE the warm-up

    for _ in range(args.warmup):
        loss = step()
    timing = rank == 0 and not args.no_kernel_timing and not args.graph
    if timing:
        ops.TIMER = ops.KernelTimer(None if args.time_all_kernels else ("gemm ", "attn", "pe_", "tokens_gather"))
    if reducer is not None:
        reducer.reset_stats()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if ops.TIMER is not None:
            ops.TIMER.active = i % max(1, args.time_every) == 0
        loss = step()
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

    rccl_stats = reducer.stats(args.steps) if reducer is not None else None     # collective calls inside: every rank
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
            "config": {"workload": f"{args.workload}: {img}x{img} {ORDER_NAME[tok]} pixel order, 16x16-pixel tokens, "
                                   f"batch {batch}/GPU, full train step (fwd+CE+bwd+clip+AdamW)",
                       "global_batch": world * batch, "parallelism": f"dp{world}", "dropout": args.dropout,
                       "hip_graph": bool(args.graph)},
            "train_gflop_per_image": round(flops_img / 1e9, 3),
            "step_tflops_per_gpu": round(value / world * flops_img / 1e12, 2),
            "step_mfma_frac": round(value / world * flops_img / 1e12 / PEAK_BF16_TFLOPS, 4),
            "final_loss": round(loss_v, 4),
        }
        if world > 1:
            out["rccl"] = dict(rccl_stats, backend=backend, world_size=dist.get_world_size())
        if kern:
            tsteps = len(range(0, args.steps, max(1, args.time_every)))     # steps whose launches carried event pairs
            traffic, traffic_build = load_traffic(args.workload)
            gemm_keys = [k for k in kern if k.startswith("gemm ")]
            fam = [k for k in gemm_keys if k.startswith("gemm gemm8p_kernel<")]
            detail = {}
            for k in gemm_keys:
                detail[k[5:]] = frac_entry(kern[k])
                t = traffic.get(k[5:], {}).get("traffic_bytes")
                if t is not None:
                    detail[k[5:]]["traffic"] = t
            if fam:
                tot = {"launches": sum(kern[k]["launches"] for k in fam), "ms_total": sum(kern[k]["ms_total"] for k in fam),
                       "work_total": sum(kern[k]["work_total"] for k in fam)}
                tot["ms_avg"] = tot["ms_total"] / tot["launches"]
                ach = tot["work_total"] / (tot["ms_total"] * 1e-3) / 1e12
                known = [k for k in fam if traffic.get(k[5:], {}).get("traffic_bytes") is not None]
                tr = None
                if known and len(known) == len(fam):
                    tr = sum(traffic[k[5:]]["traffic_bytes"] * kern[k]["launches"] for k in fam) / tot["launches"]
                out["roofline"] = {
                    "kernel": "gemm8p_kernel<*> (persistent 8-phase GEMM: every forward and dX GEMM; all instantiations, flop-weighted)",
                    "rocprof_symbol": "gemm8p_kernel", "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": tr,
                    "traffic_source": (f"profiles/traffic.json [{args.workload}] build {traffic_build}: rocprofv3 --pmc FETCH_SIZE / "
                                       "WRITE_SIZE passes, FETCH doubled per MI355X_MICROARCH.md; launch-weighted mean over "
                                       "the family") if tr is not None else None,
                    "launches": tot["launches"], "avg_launch_ms": round(tot["ms_avg"], 4),
                    "flops_per_launch": tot["work_total"] / tot["launches"], "ms_per_step": round(tot["ms_total"] / tsteps, 3)}
                worst = min(fam, key=lambda k: kern[k]["work_total"] / kern[k]["ms_total"])
                detail["worst_gemm"] = dict(detail[worst[5:]], kernel=worst[5:])
            all_g = {"launches": sum(kern[k]["launches"] for k in gemm_keys), "ms_total": sum(kern[k]["ms_total"] for k in gemm_keys),
                     "work_total": sum(kern[k]["work_total"] for k in gemm_keys)}
            all_g["ms_avg"] = all_g["ms_total"] / max(1, all_g["launches"])
            detail["all_gemms"] = frac_entry(all_g)
            if not fam and all_g["launches"]:
                # narrow models (ViT-Tiny: D = 192) have no shape the persistent kernel takes: the dominant family is then
                # every GEMM launch of the step on the older kernels (gemm256_kernel / gemm_kernel), flop-weighted
                ach = all_g["work_total"] / (all_g["ms_total"] * 1e-3) / 1e12
                out["roofline"] = {
                    "kernel": "gemm256_kernel<*> + gemm_kernel<*> (every GEMM launch of the step; no shape of this workload "
                              "is eligible for the persistent 8-phase kernel), flop-weighted",
                    "rocprof_symbol": "gemm", "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None, "traffic_source": None,
                    "launches": all_g["launches"], "avg_launch_ms": round(all_g["ms_avg"], 4),
                    "flops_per_launch": all_g["work_total"] / all_g["launches"],
                    "ms_per_step": round(all_g["ms_total"] / tsteps, 3)}
            bh = batch * N
            if "attn_fwd_kernel" in kern:
                detail["attention_fwd"] = frac_entry(kern["attn_fwd_kernel"], bytes_per_launch=bh * 4 * D * 2 + batch * heads * N * 4)
                detail["attention_fwd"]["ms_per_step"] = round(kern["attn_fwd_kernel"]["ms_total"] / tsteps, 3)
            if "attn_bwd" in kern:        # algorithmic minimum: read qkv, o, do, lse; write dqkv
                detail["attention_bwd"] = frac_entry(kern["attn_bwd"], bytes_per_launch=bh * 8 * D * 2 + batch * heads * N * 4)
                detail["attention_bwd"]["ms_per_step"] = round(kern["attn_bwd"]["ms_total"] / tsteps, 3)
            pe_in = batch * 3 * img * img * 4       # the fp32 image batch as the model receives it
            if "pe_fwd_kernel" in kern:
                detail["patch_embed_fwd"] = frac_entry(kern["pe_fwd_kernel"], bytes_per_launch=pe_in + bh * D * 2 + D * 3 * patch * 2)
            if "pe_bwd_kernel" in kern:
                detail["patch_embed_bwd"] = frac_entry(kern["pe_bwd_kernel"], bytes_per_launch=pe_in + bh * D * 2 + D * 3 * patch * 4)
            if "tokens_gather" in kern:     # two-stage patch embed: the gather alone (HBM-bound: image in, bf16 tokens out); its
                # projection and weight gradient are ordinary launches of the GEMM families above
                detail["tokens_gather"] = frac_entry(kern["tokens_gather"], bound_flops=False, bytes_per_launch=pe_in + bh * 3 * patch * 2)
                gsym = [k for k in traffic if k.startswith("tokens_gather")]      # PMC bytes per launch of the gather kernel that ran
                if len(gsym) == 1 and traffic[gsym[0]].get("traffic_bytes") is not None:
                    detail["tokens_gather"]["traffic"] = traffic[gsym[0]]["traffic_bytes"]
                    detail["tokens_gather"]["rocprof_symbol"] = gsym[0]
            detail["step"] = {"tflops": out["step_tflops_per_gpu"], "mfma_frac": out["step_mfma_frac"]}
            out["roofline_detail"] = detail
            out["kernel_timing"] = (f"HIP events around every launch of the GEMMs / attention / patch embed in {tsteps} of the "
                                    f"{args.steps} timed steps")
            out["gemm_ms_per_step"] = round(all_g["ms_total"] / tsteps, 3)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_batch or cbatch, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
