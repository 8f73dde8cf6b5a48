"""The N>1 path on CPU: world sizes 2, 4 and 8, gloo.  GradReducer + FlatGradBuffer are device-agnostic torch
plumbing (the HIP step is not involved), so the bucketed, hook-driven all-reduce is exercised here with
a small fp32 model: the reduced flat gradient must equal the average of the per-rank gradients, over two
steps (first step = synchronous reduce after the flat layout is built, later steps = hooks)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(7)                      # identical init on every rank
    m = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.ReLU(), torch.nn.Linear(40, 40), torch.nn.ReLU(),
                            torch.nn.Linear(40, 8))
    unused = torch.nn.Linear(3, 3)            # never used in forward: must stay outside the buckets
    return m, unused


def _batch(rank, step):
    g = torch.Generator().manual_seed(100 + 10 * step + rank)
    return torch.randn(16, 24, generator=g), torch.randn(16, 8, generator=g)


def _worker(rank, world, port, out):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sfcvit.training.optim import FlatGradBuffer
    from sfcvit.training.distributed import GradReducer
    model, unused = _model()
    buf = FlatGradBuffer(list(model.parameters()) + list(unused.parameters()))
    red = GradReducer(buf, bucket_bytes=4096)          # several buckets
    assert buf.grad_scale == 1.0 / world
    result = []
    for step in range(3):
        x, y = _batch(rank, step)
        buf.zero_grad()
        red.begin_step()
        ((model(x) - y) ** 2).mean().backward()
        red.finish()
        result.append((buf.flat_grad * buf.grad_scale).clone())
        assert all(p.grad is None for p in unused.parameters())
    if rank == 0:
        torch.save({"grads": result, "buckets": red.buckets, "n": buf.flat_grad.numel()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bucketed_allreduce(tmp_path, world):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    assert len(got["buckets"]) > 1
    assert got["buckets"][0][0] == 0 and got["buckets"][-1][1] == got["n"]
    for (s0, e0), (s1, e1) in zip(got["buckets"], got["buckets"][1:]):
        assert e0 == s1                                  # contiguous cover of the flat buffer
    # single-process expectation: average over the ranks' batches
    for step in range(3):
        model, _ = _model()
        ref = None
        for rank in range(world):
            model.zero_grad()
            x, y = _batch(rank, step)
            ((model(x) - y) ** 2).mean().backward()
            flat = torch.cat([torch.nn.functional.pad(p.grad.flatten(), (0, (-p.numel()) % 8)) for p in model.parameters()])
            ref = flat if ref is None else ref + flat
        ref = ref / world
        assert torch.allclose(got["grads"][step], ref, atol=1e-6), step


def test_bucket_plan():
    sys.path.insert(0, PKG)
    from sfcvit.training.distributed import GradReducer
    assert GradReducer.plan([10, 100, 1000, 8, 50], 128) == [(0, 2), (2, 3), (3, 5)]
    assert GradReducer.plan([8], 1) == [(0, 1)]
    assert GradReducer.plan([], 64) == []
    # the first bucket (last to complete in backward, its all-reduce exposed) may have its own, smaller limit
    assert GradReducer.plan([16, 16, 16, 64, 64, 64], 128, first_elems=32) == [(0, 2), (2, 4), (4, 6)]


# ---- bounded waits, the rehearsal guard, fp32 master after a resume ---------------------
def _stuck_worker(rank, world, port, out):
    """Rank 1 never reduces its second step: rank 0's wait must end in CollectiveTimeout naming the bucket."""
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SFCVIT_DIST_TIMEOUT="3")
    from sfcvit.training.distributed import CollectiveTimeout, GradReducer, dist_timeout
    from sfcvit.training.optim import FlatGradBuffer
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=dist_timeout())
    model, _ = _model()
    buf = FlatGradBuffer(list(model.parameters()))
    red = GradReducer(buf, bucket_bytes=4096)
    msg = ""
    for step in range(2):
        x, y = _batch(rank, step)
        buf.zero_grad()
        red.begin_step()
        if step == 1 and rank == 1:
            import time
            time.sleep(8)                       # alive, but not taking part (the hooks of its backward would launch its share)
            break
        ((model(x) - y) ** 2).mean().backward()
        try:
            red.finish()
        except CollectiveTimeout as e:
            msg = str(e)
            break
    if rank == 0:
        torch.save({"msg": msg}, out)


@pytest.mark.timeout(120)
def test_a_stuck_collective_raises_with_the_bucket_index(tmp_path):
    out = str(tmp_path / "r0.pt")
    try:
        mp.spawn(_stuck_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    except Exception:
        pass                                    # teardown of a half-finished group may complain; rank 0's report is what counts
    msg = torch.load(out)["msg"]
    assert "all-reduce of gradient bucket" in msg and "of step 1" in msg and "within 3 s" in msg, msg


def test_more_than_two_gloo_ranks_on_one_device_are_refused():
    sys.path.insert(0, PKG)
    from sfcvit.training.distributed import check_rehearsal_layout
    check_rehearsal_layout("gloo", 2, 2)
    check_rehearsal_layout("nccl", 8, 1)
    check_rehearsal_layout("gloo", 4, 1)        # CPU tensors / one rank per device: fine
    with pytest.raises(RuntimeError, match="4 gloo ranks on one GPU"):
        check_rehearsal_layout("gloo", 4, 4)


class _MasterBuffer:
    """FlatGradBuffer with fp32 master weights as FusedAdamW keeps them (that class needs a GPU): bf16 parameters and
    gradients, master = what a checkpoint restored."""


def _resume_worker(rank, world, port, out):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sfcvit.training.distributed import GradReducer
    from sfcvit.training.optim import FlatGradBuffer

    class Buf(FlatGradBuffer):
        def _built(self):
            self.master = self.flat_param.float()

    torch.manual_seed(7)
    model = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.ReLU(), torch.nn.Linear(40, 8)).to(torch.bfloat16)
    buf = Buf(list(model.parameters()))
    buf._build(list(model.parameters()))
    # "--resume": rank 0 restores fp32 master weights that carry bits below bf16's; rank 1 holds something else entirely
    g = torch.Generator().manual_seed(5)
    ckpt = torch.randn(buf.master.shape, generator=g) * 0.1
    buf.master.copy_(ckpt if rank == 0 else torch.zeros_like(ckpt))
    buf.flat_param.copy_(buf.master)
    red = GradReducer(buf, bucket_bytes=1024)
    x, y = _batch(rank, 0)
    buf.zero_grad()
    red.begin_step()
    ((model(x.bfloat16()).float() - y) ** 2).mean().backward()
    red.finish()                                 # first reduced step: installs the hooks, broadcasts rank 0's state
    torch.save({"master": buf.master.clone(), "param": buf.flat_param.clone(), "ckpt": ckpt, "grad": buf.flat_grad.clone()}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_first_reduced_step_keeps_the_checkpointed_fp32_master(tmp_path):
    """on a multi-rank resume the first step's broadcast must carry the fp32 master, not the bf16-rounded
    parameters -- afterwards every rank's master is bit-identical to the checkpoint and its bf16 parameters are the
    rounded master.  (bf16 gradients reduce over gloo as well.)"""
    out = str(tmp_path / "rank")
    mp.spawn(_resume_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for r in (r0, r1):
        assert torch.equal(r["master"], r0["ckpt"])
        assert torch.equal(r["param"], r0["ckpt"].to(torch.bfloat16))
    assert (r0["ckpt"] - r0["ckpt"].to(torch.bfloat16).float()).abs().max() > 0     # the checkpoint does carry sub-bf16 bits
    assert torch.equal(r0["grad"], r1["grad"]) and r0["grad"].dtype == torch.bfloat16 and r0["grad"].abs().sum() > 0


# ---- precision of the bf16 gradient SUM at world 8 (VERDICT r3 #5b) ------------------------------------------------
def _rank_grad(rank, n):
    """A per-rank bf16 gradient: a component all ranks share (the signal) plus a per-rank one twice as large (batch noise),
    magnitudes spread over three decades as a model's gradients are."""
    g = torch.Generator().manual_seed(900)
    scale = torch.pow(10.0, -3 * torch.rand(n, generator=g))
    common = torch.randn(n, generator=g)
    own = torch.randn(n, generator=torch.Generator().manual_seed(901 + rank))
    return ((common + 2 * own) * scale * 1e-2).to(torch.bfloat16)


def _bf16_sum_worker(rank, world, port, out):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sfcvit.training.distributed import GradReducer
    from sfcvit.training.optim import FlatGradBuffer
    n = 1 << 17
    res = {}
    for name, dt in (("bf16", None), ("fp32", torch.float32)):
        ps = [torch.nn.Parameter(torch.zeros(n // 8, dtype=torch.bfloat16)) for _ in range(8)]
        buf = FlatGradBuffer(ps)
        red = GradReducer(buf, bucket_bytes=64 << 10, reduce_dtype=dt)
        for step in range(2):                      # step 0 builds the flat layout, step 1 runs from the hooks
            buf.zero_grad()
            red.begin_step()
            (torch.cat(ps) * _rank_grad(rank, n)).sum().backward()
            red.finish()
        assert buf.flat_grad.dtype == torch.bfloat16 and len(red.buckets) > 1
        res[name] = buf.flat_grad[:n].clone()
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_bf16_gradient_sum_at_world_8_is_bounded_against_an_fp32_sum(tmp_path):
    """The reducer sums bf16 gradients across the ranks in bf16 (distributed.py; half the xGMI bytes of fp32).  With 8
    addends that is up to 7 roundings per element instead of one.  Measured here against the exact (fp64) sum of the same
    8 per-rank bf16 gradients: (a) gloo's bf16 all-reduce as the reducer runs it, (b) a ring's hop-by-hop bf16 running sum
    in the worst rank order (what RCCL's ring reduce-scatter does), (c) the reducer's fp32 option (reduce_dtype /
    SFCVIT_DDP_FP32=1), which must be as good as one final rounding.  The bound for (a) and (b) is a relative L2 error of
    1e-2 and a norm error of 2e-3 -- an order of magnitude inside the 2.5 % gradient-norm envelope the single-GPU training
    step is held to (tests/test_parity_gpu.py) -- so bf16 stays the default."""
    world, n = 8, 1 << 17
    out = str(tmp_path / "r0.pt")
    mp.spawn(_bf16_sum_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    parts = [_rank_grad(r, n) for r in range(world)]
    exact = torch.stack([p.double() for p in parts]).sum(0)
    ring = parts[0].clone()
    for p in parts[1:]:
        ring = (ring.float() + p.float()).to(torch.bfloat16)        # one bf16 rounding per hop
    once = exact.to(torch.bfloat16)                                  # the best any bf16 result can be

    def err(x):
        d = x.double() - exact
        return float(d.norm() / exact.norm()), abs(float(x.double().norm() / exact.norm()) - 1)
    e_gloo, e_ring, e_fp32, e_once = err(got["bf16"]), err(ring), err(got["fp32"]), err(once)
    print(f"relative L2 error / norm error of the 8-rank sum: gloo bf16 {e_gloo}, ring-order bf16 {e_ring}, "
          f"fp32 option {e_fp32}, one rounding {e_once}")
    assert e_gloo[0] <= 1e-2 and e_gloo[1] <= 2e-3
    assert e_ring[0] <= 1e-2 and e_ring[1] <= 2e-3
    assert e_fp32[0] <= 1.05 * e_once[0] + 1e-6                     # fp32 sum, rounded once
    assert e_once[0] < e_ring[0]                                    # the measurement can tell the two apart
