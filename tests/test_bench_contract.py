"""bench.py's output contract (one JSON line per run; the driver parses it) and its N > 1 code path.

CPU: `--cpu-only` (BASELINE config 1: the CPU restatement timed on the host cores) prints the contract's keys.
GPU: a short single-GPU run carries `roofline` and `roofline_detail`; two ranks launched the way the driver launches
them (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ... bench.py --gpus 2`)
run the data-parallel path end to end -- on this one-GPU box over gloo with both ranks on card 0
(SFCVIT_DIST_BACKEND / SFCVIT_FORCE_DEVICE, rehearsal knobs the driver's RCCL runs do not set)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ, **(env or {}))
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_cpu_only_line():
    d = _run([sys.executable, "bench.py", "--cpu-only", "--cpu-steps", "1"])
    assert KEYS <= set(d) and d["n_gpus"] == 0 and d["unit"] == "images/s" and d["value"] > 0
    assert d["vs_baseline"] is None and d["higher_is_better"] is True
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and "sample" in cb and cb["value"] == pytest.approx(d["value"], rel=1e-3)
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.gpu
def test_single_gpu_line_has_roofline():
    d = _run([sys.executable, "bench.py", "--workload", "vit_tiny16_32_hilbert", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["dtype"] == "bf16" and d["data"].startswith("synthetic")
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] in ("TFLOP/s", "GB/s") and 0 < r["frac"] < 1
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-2, abs=1e-4)      # both are rounded in the line
    assert "attention_fwd" in d["roofline_detail"] and "workload" in d["config"]


@pytest.mark.gpu
def test_two_rank_launch_runs_the_data_parallel_path():
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    if backend == "gloo":
        env.update(SFCVIT_DIST_BACKEND="gloo", SFCVIT_FORCE_DEVICE="0")
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29533", "bench.py", "--gpus", "2", "--workload", "vit_tiny16_32_hilbert", "--steps", "3",
              "--warmup", "2", "--no-cpu-baseline"], env=env)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 2 * 256 and d["config"]["parallelism"] == "dp2"
    rc = d["rccl"]
    assert rc["world_size"] == 2 and rc["backend"] == backend and rc["bytes_per_step"] > 0 and rc["buckets"] >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_entry_script_trains_checkpoints_and_resumes(tmp_path, graph):
    """The repository's main.py (structure, defaults and printed line of the reference's main.py:255-354) as a
    subprocess on synthetic CIFAR-shaped batches: one epoch of the default model (hierarchical Morton tokenizer, 768
    wide, 8 layers, 4 heads = head dim 192), eagerly and with --graph; the checkpoint carries the reference's keys and
    --resume continues from it."""
    import re
    import torch
    pkg = os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd")
    base = [sys.executable, os.path.join(pkg, "main.py"), "--synthetic", "--train-size", "1024", "--test-size", "512",
            "--batch-size", "256", "--checkpoint-dir", str(tmp_path)] + (["--graph"] if graph else [])
    r = subprocess.run(base + ["--epochs", "1"], cwd=pkg, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("Epoch 1/1")]
    assert len(line) == 1 and re.search(r"Train Loss: \d+\.\d{4}, Train Acc: \d\.\d{4} \| Test Loss: \d+\.\d{4}, Test Acc: \d\.\d{4}", line[0])
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint_hier_morton.pt"), map_location="cpu", weights_only=True)
    assert {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "train_loss", "train_acc", "test_loss",
            "test_acc"} <= set(ck)                                           # main.py:345-354
    assert ck["epoch"] == 0 and "patch_embed.fusion.weight" in ck["model_state_dict"]
    r2 = subprocess.run(base + ["--epochs", "2", "--resume", os.path.join(str(tmp_path), "checkpoint_hier_morton.pt")],
                        cwd=pkg, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    out = [ln for ln in r2.stdout.splitlines() if ln.startswith("Epoch")]
    assert len(out) == 1 and out[0].startswith("Epoch 2/2")                  # epoch 1 came from the checkpoint
