"""Data parallelism over RCCL on real devices: 2 fresh ranks (one process per GPU, backend "nccl" = RCCL on ROCm),
the HIP training step with the bucketed, hook-driven gradient all-reduce (sfcvit.training.GradReducer), against ONE
rank taking the same steps on the concatenated batch.  Needs >= 2 visible GPUs: skipped on the one-GPU box, runs
wherever the driver has a multi-GPU node."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd")
STEPS, LR = 3, 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup():
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import formula, vit_oracle
    from oracle.cases import MODEL_CASES
    from test_host_cpu import build_model
    cfg, _ = MODEL_CASES["hilbert32_1d"]
    model = build_model(cfg)
    model.load_state_dict(vit_oracle.formula_state(cfg), strict=True)
    batch = 8
    x = formula.image_batch(batch, cfg.in_channels, cfg.img_size, cfg.img_size)
    tgt = formula.soft_targets(batch, cfg.num_classes)
    return model, x, tgt


def _run(model, x, tgt, dev, reducer_world, graph=False):
    from sfcvit.training import FusedAdamW, GradReducer, GraphedTrainStep, train_step
    model = model.to(dev, dtype=torch.bfloat16).eval()          # dropout off: the two runs must see the same function
    opt = FusedAdamW(model.parameters(), lr=LR, weight_decay=5e-2, max_grad_norm=1.0)
    red = GradReducer(opt, bucket_bytes=1 << 16, overlap=not graph) if reducer_world else None   # small buckets: several collectives in flight
    if graph:     # forward + backward and the optimizer step replay from two hipGraphs, the collectives run between them
        step = GraphedTrainStep(model, x.to(dev), tgt.to(dev), opt, warmup=2, reducer=red)
        losses = [float(step()) for _ in range(STEPS)]
        step.close()
    else:
        losses = [float(train_step(model, x.to(dev), tgt.to(dev), opt, reducer=red)) for _ in range(STEPS)]
    names = {id(p): k for k, p in model.named_parameters()}
    master = {names[id(p)]: opt.master[o:o + p.numel()].view(p.shape).cpu() for p, o in zip(opt.active, opt.offsets)}
    return losses, master, (len(red.buckets) if red else 0)


def _worker(rank, world, port, out, backend="nccl"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from sfcvit.training.distributed import dist_timeout
    graph = backend.endswith("+graph")
    backend = backend.split("+")[0]
    card = rank if backend == "nccl" else 0                  # gloo rehearsal: both ranks share card 0
    torch.cuda.set_device(card)
    dev = torch.device("cuda", card)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=dist_timeout())
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=dist_timeout())
    model, x, tgt = _setup()
    per = x.shape[0] // world
    losses, master, nb = _run(model, x[rank * per:(rank + 1) * per], tgt[rank * per:(rank + 1) * per], dev, world, graph)
    torch.save({"losses": losses, "master": master, "buckets": nb}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("backend", ["nccl", "nccl+graph", "gloo", "gloo+graph"])
def test_two_ranks_match_one_rank_on_the_concatenated_batch(tmp_path, backend):
    """backend nccl: two GPUs over RCCL (skipped on a one-GPU box).  backend gloo: the rehearsal that always runs -- two
    ranks of the HIP step on ONE card, gradients reduced through the host (GradReducer's host-staged gloo mode), so the
    N > 1 step has a correctness check that is never skipped.  "+graph": data parallelism composed with
    graph replay -- GraphedTrainStep(reducer=GradReducer(overlap=False)): forward + backward and the optimizer step replay
    from two hipGraphs, the gradient all-reduce runs between them."""
    if backend.startswith("nccl") and torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL over xGMI)")
    import torch.multiprocessing as mp
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(2, _free_port(), out, backend), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["buckets"] > 1
    for k in r0["master"]:
        assert torch.equal(r0["master"][k], r1["master"][k]), k          # replicas stay bit-identical
    model, x, tgt = _setup()
    init = {k: v.detach().clone().bfloat16().float() for k, v in model.named_parameters()}
    losses, master, _ = _run(model, x, tgt, torch.device("cuda", 0), 0)
    for s in range(STEPS):      # the global loss is the mean of the two half-batch losses
        assert abs(0.5 * (r0["losses"][s] + r1["losses"][s]) - losses[s]) <= 1e-2 * abs(losses[s]) + 2e-3
    confident = total = 0
    for k, w in master.items():
        upd, dp = (w - init[k]).flatten(), (r0["master"][k] - init[k]).flatten()
        assert (upd - dp).abs().max() <= 2.1 * LR * STEPS, k
        mask = upd.abs() >= 0.8 * LR * STEPS                              # elements with a consistent gradient sign (see
        confident += int(mask.sum())                                      # test_train_steps_match_oracle_and_reference)
        total += mask.numel()
        if int(mask.sum()) >= 8:
            assert float((torch.sign(upd[mask]) == torch.sign(dp[mask])).float().mean()) >= 0.9, k
    assert confident >= 0.1 * total
