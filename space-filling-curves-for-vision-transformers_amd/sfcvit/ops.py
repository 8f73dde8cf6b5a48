"""Thin, allocation-only wrappers over the C ABI (include/sfcvit.h).

Every function here takes torch CUDA tensors, checks dtype / contiguity, allocates
outputs with torch's caching allocator and launches the HIP kernels on torch's
current stream.  There is no CPU path: a CPU tensor raises.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import lib, check

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
_BF16 = torch.bfloat16


class KernelTimer:
    """Optional per-launch timing with HIP events on the launching stream (bench.py's roofline
    leg).  Set `ops.TIMER = KernelTimer()`; every wrapper below then brackets its launch."""

    def __init__(self, only_prefix=None):
        self.records = {}
        self.only_prefix = only_prefix      # time only keys with this prefix (an event pair costs ~1 us of stream time)
        self.active = True                  # bench.py switches it per step: every launch of every 4th timed step is timed

    def launch(self, key, work, fn):
        if not self.active or (self.only_prefix and not key.startswith(self.only_prefix)):
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        if key.startswith("gemm "):         # key GEMM timings by the kernel symbol the library chose (as rocprofv3 names it)
            key = "gemm " + last_gemm_kernel()
        self.records.setdefault(key, []).append((e0, e1, work))
        return out

    def summary(self):
        """key -> dict(launches, ms_total, ms_avg, work_total); synchronises."""
        torch.cuda.synchronize()
        out = {}
        for key, recs in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in recs)
            out[key] = {"launches": len(recs), "ms_total": ms, "ms_avg": ms / len(recs),
                        "work_total": float(sum(w for _, _, w in recs))}
        return out


TIMER = None
KERNEL_LOG = None     # tests: set to a list and every GEMM / attention launch appends the kernel symbol the library chose


def last_gemm_kernel():
    buf = ctypes.create_string_buffer(96)
    lib.sfcvit_last_gemm_kernel(buf, 96)
    return buf.value.decode()


def last_attn_kernel():
    buf = ctypes.create_string_buffer(96)
    lib.sfcvit_last_attn_kernel(buf, 96)
    return buf.value.decode()


def last_rowwise_kernel():
    buf = ctypes.create_string_buffer(96)
    lib.sfcvit_last_rowwise_kernel(buf, 96)
    return buf.value.decode()


def _launch(key, work, fn):
    out = fn() if TIMER is None else TIMER.launch(key, work, fn)
    if KERNEL_LOG is not None:
        if key.startswith("gemm "):
            KERNEL_LOG.append(last_gemm_kernel())
        elif key.startswith("attn"):
            KERNEL_LOG.append(last_attn_kernel())
    return out


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _cur_dev(t, name):
    """One process per GPU: kernels are launched on the CURRENT device's current stream, and the library keeps
    per-(device, stream) state (tile-queue counters).  A tensor of another device would be dereferenced through the
    wrong context, so it is refused here rather than faulting there."""
    if t.device.index != torch.cuda.current_device():
        raise _lib.SfcvitError(f"{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                               "wrap the call in torch.cuda.device(tensor.device) (one process per GPU is the supported layout)")


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _need(t, dtype, name, dims=None):
    if not t.is_cuda:
        raise _lib.SfcvitError(f"{name}: the HIP path needs a CUDA (ROCm) tensor, got device {t.device}; "
                               "there is no CPU fallback")
    _cur_dev(t, name)
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if dims is not None and t.dim() != dims:
        raise ValueError(f"{name}: expected {dims}-D, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _rows2d(t, dtype, name):
    """2-D, unit inner stride, arbitrary row stride (views of packed buffers are fine)."""
    if not t.is_cuda:
        raise _lib.SfcvitError(f"{name}: the HIP path needs a CUDA (ROCm) tensor; there is no CPU fallback")
    _cur_dev(t, name)
    if t.dtype != dtype or t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected 2-D {dtype} with unit inner stride, got {t.dtype} {tuple(t.shape)} {t.stride()}")
    return t


# ----------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------
def auto_splitk(M, N, K):
    """Split K when the output has too few 128 x 128 tiles to fill 256 CUs at 2 workgroups each
    (weight-gradient shapes: 36-144 tiles, K = batch * tokens).  The split is chosen so that
    tiles * splits fills whole rounds of the 512 workgroup slots (864 workgroups = 1.69 rounds cost
    15 % against 1008 = 1.97), each k-range keeps >= 896 deep, and -- for outputs under 64 tiles,
    where sfcvit_gemm gives every XCD its own k-ranges -- is a multiple of 8."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if K < 2048:
        return 1
    if M % 256 == 0 and N % 256 == 0 and (M // 256) * (N // 256) <= 128:      # any K: the last K % 128 rows are one more slab
        # weight-gradient form of the 8-phase kernel: one workgroup per CU, tiles256 x splits <= 256
        s = min(64, 256 // ((M // 256) * (N // 256)))
        if s >= 2 and K // s >= 512:
            return s
    if tiles >= 256:
        return 1
    step = 8 if tiles < 64 else 1
    best, best_eff = 1, tiles / 512.0 if tiles < 512 else 1.0
    for s in range(max(2, step), 65, step):
        if K // s < 896:
            break
        wgs = tiles * s
        eff = wgs / (-(-wgs // 512) * 512)
        if eff > best_eff + 0.02:
            best, best_eff = s, eff
        if best_eff >= 0.93:
            break
    return best


# Device-resident step state (include/sfcvit.h, sfcvit_step_advance): None = seeds and Adam's step count live on the host
# (the default, as in the reference); a CUDA int32[8] tensor = every dropout site adds the device seed offset and AdamW
# reads lr / bias corrections from the device, so a whole training step can replay from one hipGraph
# (sfcvit.training.GraphedTrainStep).
STEP_STATE = None


def _seed_off():
    return ctypes.c_void_p(STEP_STATE.data_ptr()) if STEP_STATE is not None else None


_site = 0     # device-state mode: dropout sites of the current step drawn so far


def step_advance(state, beta1, beta2, seed_base):
    global _site
    _site = 0
    check(lib.sfcvit_step_advance(_p(state), beta1, beta2, seed_base & 0xFFFFFFFF, _stream()), "sfcvit_step_advance")


# Deferred reductions (include/sfcvit.h: sfcvit_reduce_defer / _flush).  Bias, LayerNorm-parameter and similar gradients are
# a main kernel + a 5-us fixed-order reduction over its partial rows; inside a backward pass nobody reads them before the pass
# ends when they are written into gradient slots (functional._slot marks those views), so their reductions are queued and
# run as ONE launch from an autograd end-of-pass callback (66 launches per ViT-B step become one).  A reducer that ships
# gradients mid-pass calls flush_deferred(end=False) first.  SFCVIT_DEFER_REDUCE=0 switches it off (A/B, tests).
import os as _os
DEFER_REDUCES = _os.environ.get("SFCVIT_DEFER_REDUCE", "1") != "0"
_defer = {"task": -1, "keep": []}
# autograd's id of the running backward pass (-1 outside one); a torch without it simply never defers / batches
graph_task_id = getattr(torch._C, "_current_graph_task_id", lambda: -1)


def flush_deferred(end=True):
    """Launch every queued reduction on the current stream (end=False: the backward pass goes on deferring)."""
    if lib.sfcvit_reduce_pending():
        check(lib.sfcvit_reduce_flush(_stream()), "sfcvit_reduce_flush")
    _defer["keep"].clear()
    if end:
        _defer["task"] = -1


class _Deferring:
    """with _Deferring(outputs, workspaces): the C calls inside queue their final reductions iff every output is a
    gradient slot and a backward pass is running; the workspaces then stay alive until the flush."""

    def __init__(self, outs, keep):
        self.on, self.keep = False, keep
        if DEFER_REDUCES and outs and all(getattr(t, "_sfcvit_deferrable", False) for t in outs):
            task = graph_task_id()
            if task >= 0:
                if _defer["task"] != task:
                    if lib.sfcvit_reduce_pending():      # a pass that died with reductions queued
                        lib.sfcvit_reduce_discard()
                    _defer["keep"].clear()
                    _defer["task"] = task
                    torch.autograd.Variable._execution_engine.queue_callback(flush_deferred)
                self.on = True

    def __enter__(self):
        if self.on:
            lib.sfcvit_reduce_defer(1)
        return self

    def __exit__(self, *exc):
        if self.on:
            lib.sfcvit_reduce_defer(0)
            _defer["keep"].extend(self.keep)
        return False


def next_seed():
    """32-bit dropout seed drawn on the host from torch's default CPU generator: reproducible under
    torch.manual_seed, no device synchronisation (kernels receive it as a plain argument)."""
    global _site
    if STEP_STATE is not None:
        # device-state mode: the by-value seed only tells the sites of one step apart (it is frozen into a captured graph);
        # what changes from step to step is the device offset the kernels add to it
        _site += 1
        return ((_site * 0x9E3779B1) & 0x7FFFFFFF) | 1
    return int(torch.randint(0, 0x7FFFFFFF, (), dtype=torch.int64)) * 2 + 1


def gemm(a, b, *, a_kmajor=False, b_kmajor=False, bias=None, act=ACT_NONE, residual=None,
         aux_in=None, dact=ACT_NONE, want_aux=False, out_f32=False, splitk=None, force_generic=False,
         dropout_p=0.0, dropout_seed=0, dact_scale=1.0, out=None, row_offset=0, colsum=None, actmask=None):
    """C[M,N] = epilogue(sum_k A(m,k) B(n,k)).  a: [M,K] (or [K,M] if a_kmajor),
    b: [N,K] (or [K,N] if b_kmajor), bf16.  Returns C (and the pre-activation if want_aux; and the column sums of C
    if `colsum` is True (fp32 [N]) or a bf16 [N] tensor to write them into, e.g. a bias-gradient slot).
    actmask: uint8 [M, N/8] bit matrix (bit n & 7 of byte n >> 3 <-> C[m, n] > 0): written with act = RELU, optionally
    read in place of aux_in with dact = RELU (sfcvit_gemm_args.actmask)."""
    _rows2d(a, _BF16, "gemm a")
    _rows2d(b, _BF16, "gemm b")
    M, K = (a.shape[1], a.shape[0]) if a_kmajor else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if b_kmajor else (b.shape[0], b.shape[1])
    if K != Kb:
        raise ValueError(f"gemm: contraction mismatch {K} vs {Kb}")
    if out is not None:
        c = _rows2d(out, torch.float32 if out_f32 else _BF16, "gemm out")
        if c.shape != (M, N):
            raise ValueError(f"gemm out: expected {(M, N)}, got {tuple(c.shape)}")
    else:
        c = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else _BF16)
    aux = torch.empty((M, N), device=a.device, dtype=_BF16) if want_aux else None
    args = _lib.GemmArgs()
    args.a, args.b, args.c = a.data_ptr(), b.data_ptr(), c.data_ptr()
    args.bias = _need(bias, _BF16, "gemm bias", 1).data_ptr() if bias is not None else None
    args.M, args.N, args.K = M, N, K
    args.lda, args.ldb, args.ldc = a.stride(0), b.stride(0), c.stride(0)
    if residual is not None:
        _rows2d(residual, _BF16, "gemm residual")
        args.residual, args.ldr = residual.data_ptr(), residual.stride(0)
    ldaux = N
    if aux_in is not None:
        _rows2d(aux_in, _BF16, "gemm aux_in")
        args.aux_in, ldaux = aux_in.data_ptr(), aux_in.stride(0)
    if aux is not None:
        args.aux_out = aux.data_ptr()
    args.ldaux = ldaux
    args.a_kmajor, args.b_kmajor = int(a_kmajor), int(b_kmajor)
    args.act, args.dact, args.c_is_f32 = act, dact, int(out_f32)
    args.dropout_p, args.dropout_seed, args.dact_scale = dropout_p, dropout_seed, dact_scale
    if dropout_p > 0.0 and STEP_STATE is not None:
        args.seed_off = STEP_STATE.data_ptr()
    args.row_offset = row_offset
    if actmask is not None:
        if actmask.dtype != torch.uint8 or actmask.dim() != 2 or actmask.shape[0] != M or actmask.shape[1] * 8 < N \
                or actmask.stride(1) != 1 or actmask.device != a.device:
            raise ValueError(f"gemm actmask: expected uint8 [{M}, >= {N // 8}] on {a.device}, got {actmask.dtype} {tuple(actmask.shape)}")
        args.actmask, args.ld_actmask = actmask.data_ptr(), actmask.stride(0)
    plain = (bias is None and residual is None and aux_in is None and not want_aux and act == 0 and dact == 0
             and dropout_p == 0.0)
    if splitk is None:
        splitk = auto_splitk(M, N, K) if plain else 1
    ws = None
    if splitk > 1:
        nbytes = lib.sfcvit_gemm_workspace(M, N, splitk)
        ws = torch.empty(nbytes, device=a.device, dtype=torch.uint8)
        args.workspace, args.workspace_bytes = ws.data_ptr(), nbytes
    args.splitk = splitk
    args.force_generic = int(force_generic)
    cs = None
    if colsum is not None and colsum is not False:
        cs = torch.empty(N, device=a.device, dtype=torch.float32) if colsum is True else colsum
        if cs.numel() != N or not cs.is_contiguous() or cs.dtype not in (torch.float32, _BF16):
            raise ValueError("gemm colsum: contiguous fp32 or bf16 [N] expected")
        nbytes = lib.sfcvit_gemm_colsum_workspace(M, N)
        ws = torch.empty(nbytes, device=a.device, dtype=torch.uint8)
        args.workspace, args.workspace_bytes = ws.data_ptr(), nbytes
        args.colsum_out, args.colsum_bf16 = cs.data_ptr(), int(cs.dtype == _BF16)
    key = "gemm %s" % {(False, False): "y=x.W^T (k-contig, k-contig)", (False, True): "dx=dy.W (k-contig, k-major)",
                       (True, True): "dW=dy^T.x (k-major, k-major)", (True, False): "(k-major, k-contig)"}[(a_kmajor, b_kmajor)]
    with _Deferring([colsum] if cs is not None and colsum is not True else [], [ws] if cs is not None else []):
        check(_launch(key, 2.0 * M * N * K, lambda: lib.sfcvit_gemm(ctypes.byref(args), _stream())), "sfcvit_gemm")
    if cs is not None:
        return (c, aux, cs) if want_aux else (c, cs)
    return (c, aux) if want_aux else c


def transpose(x):
    """[R, C] bf16 -> contiguous [C, R]."""
    _rows2d(x, _BF16, "transpose x")
    R, C = x.shape
    out = torch.empty((C, R), device=x.device, dtype=_BF16)
    check(lib.sfcvit_transpose(_p(x), R, C, x.stride(0), _p(out), R, _stream()), "sfcvit_transpose")
    return out


def transpose_batched(src_flat, dst_flat, tiles, n_tiles):
    """Every matrix of a flat bf16 buffer transposed in one launch; `tiles` = device uint8 tensor holding n_tiles
    sfcvit_transpose_tile records (FlatGradBuffer.transposed)."""
    _need(src_flat, _BF16, "transpose_batched src", 1)
    _need(dst_flat, _BF16, "transpose_batched dst", 1)
    if tiles.dtype != torch.uint8 or not tiles.is_cuda or tiles.numel() != 32 * n_tiles:
        raise ValueError("transpose_batched tiles: device uint8 tensor of 32 bytes per tile expected")
    check(lib.sfcvit_transpose_batched(_p(src_flat), _p(dst_flat), _p(tiles), n_tiles, _stream()), "sfcvit_transpose_batched")


def gemm_dx(dy, w, **kw):
    """dX[M, in] = dY[M, out] . W[out, in] (+ epilogue).  When the shape is eligible for the LDS-DMA
    kernel, W is transposed once (a few MB) so that both operands are k-contiguous; otherwise W is
    read k-major in place by the generic kernel."""
    M, K = dy.shape
    N = w.shape[1]
    # the persistent 8-phase kernel takes any M >= 192 (a last row tile that overlaps its predecessor); the LDS-DMA
    # ring kernel before it needs whole 256-row tiles
    if N % 128 == 0 and K % 32 == 0 and (M % 256 == 0 or (M >= 192 and N % 256 == 0 and K % 128 == 0 and K >= 256)):
        slot = getattr(w, "_sfcvit_slot", None)          # a parameter of a flat buffer: W^T from this backward pass's batch
        wt = slot[0].transposed(w) if slot is not None else None
        return gemm(dy, wt if wt is not None else transpose(w), **kw)
    return gemm(dy, w, b_kmajor=True, **kw)


def colsum(x, out=None):
    """Column sums [N]: fp32 (new tensor), or written as bf16 into `out` (e.g. a view of a flat gradient buffer)."""
    _rows2d(x, _BF16, "colsum x")
    if out is None:
        out = torch.empty(x.shape[1], device=x.device, dtype=torch.float32)
    elif out.dtype != _BF16 or out.numel() != x.shape[1] or not out.is_contiguous():
        raise ValueError("colsum out: contiguous bf16 [N] expected")
    nbytes = lib.sfcvit_colsum_workspace(x.shape[0], x.shape[1])
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    with _Deferring([out], [ws]):
        check(lib.sfcvit_colsum(_p(x), x.shape[0], x.shape[1], x.stride(0), _p(out), int(out.dtype == _BF16), _p(ws), nbytes,
                                _stream()), "sfcvit_colsum")
    return out


# ----------------------------------------------------------------------------
# LayerNorm
# ----------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps=1e-5):
    _need(x, _BF16, "layernorm x", 2)
    M, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    check(lib.sfcvit_layernorm_fwd(_p(x), _p(_need(gamma, _BF16, "gamma", 1)), _p(_need(beta, _BF16, "beta", 1)),
                                   _p(y), _p(mean), _p(rstd), M, D, eps, _stream()), "sfcvit_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dx_add=None, drop_p=0.0, drop_seed=0, want_colsum=False, grad_out=None):
    """-> dx, dgamma, dbeta [, dx_drop when drop_p > 0] [, colsum of the outgoing gradient (dx_drop if
    drop_p > 0 else dx) when want_colsum].  dgamma / dbeta / colsum are fp32 [D], or -- with
    grad_out = (dgamma, dbeta, colsum-or-None) bf16 [D] tensors, e.g. views of a flat gradient buffer (None entries
    are allocated) -- written as bf16 in place."""
    _need(dy, _BF16, "layernorm dy", 2)
    _need(x, _BF16, "layernorm x", 2)
    M, D = x.shape
    dx = torch.empty_like(x)
    dx_drop = torch.empty_like(x) if drop_p > 0 else None
    gdt = torch.float32 if grad_out is None else _BF16
    given = (None, None, None) if grad_out is None else grad_out
    for t in given:
        if t is not None and (t.dtype != _BF16 or t.numel() != D or not t.is_contiguous()):
            raise ValueError("layernorm_bwd grad_out: contiguous bf16 [D] tensors expected")
    dg = given[0] if given[0] is not None else torch.empty(D, device=x.device, dtype=gdt)
    db = given[1] if given[1] is not None else torch.empty(D, device=x.device, dtype=gdt)
    ws = torch.empty(lib.sfcvit_layernorm_bwd_ws(M, D), device=x.device, dtype=torch.uint8)
    if dx_add is not None:
        _need(dx_add, _BF16, "layernorm dx_add", 2)
    dcol = None
    if want_colsum:
        dcol = given[2] if given[2] is not None else torch.empty(D, device=x.device, dtype=gdt)
    with _Deferring([dg, db] + ([dcol] if dcol is not None else []), [ws]):
        check(lib.sfcvit_layernorm_bwd_drop(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx_add), _p(dx), _p(dx_drop),
                                            drop_p, drop_seed, _seed_off() if drop_p > 0 else None, _p(dg), _p(db), _p(dcol),
                                            int(gdt == _BF16), M, D, _p(ws),
                                            _stream()),
              "sfcvit_layernorm_bwd")
    if KERNEL_LOG is not None:
        KERNEL_LOG.append(last_rowwise_kernel())
    out = (dx, dg, db) + ((dx_drop,) if drop_p > 0 else ()) + ((dcol,) if want_colsum else ())
    return out


# ----------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------
SUPPORTED_HEAD_DIMS = (64, 128, 192, 256)


def _attn_dims(D3, n_heads):
    """(D, head dim) of a packed q|k|v projection; the kernels exist for head dims 64 (any N) and 128 / 192 / 256
    (whole-sequence kernels: N <= 288 / 192 / 128)."""
    if D3 % 3 or (D3 // 3) % n_heads:
        raise ValueError(f"attention: packed width {D3} is not 3 * n_heads * head_dim for n_heads = {n_heads}")
    D = D3 // 3
    hd = D // n_heads
    if hd not in SUPPORTED_HEAD_DIMS:
        raise ValueError(f"attention: head dim {hd} (= {D} / {n_heads}) is not supported; supported: {SUPPORTED_HEAD_DIMS}")
    return D, hd


def padded_head_dim(hd):
    """The kernel head dim a model head dim runs on: itself when supported, else the next supported one (zero-padded
    q / k / v columns change neither q k^T nor p v; the softmax scale stays 1 / sqrt(hd)); None above the largest."""
    for s in sorted(SUPPORTED_HEAD_DIMS):
        if hd <= s:
            return s
    return None


def attention_fwd(qkv, n_heads, dropout_p=0.0, dropout_seed=0, scale=None):
    """qkv [B, N, 3*D] bf16 -> out [B, N, D] bf16, lse [B, H, N] fp32.  scale: softmax scale, default 1 / sqrt(head dim)."""
    _need(qkv, _BF16, "attention qkv", 3)
    B, N, D3 = qkv.shape
    D, hd = _attn_dims(D3, n_heads)
    out = torch.empty((B, N, D), device=qkv.device, dtype=_BF16)
    lse = torch.empty((B, n_heads, N), device=qkv.device, dtype=torch.float32)
    a = _lib.AttnArgs()
    a.qkv, a.out, a.lse = qkv.data_ptr(), out.data_ptr(), lse.data_ptr()
    a.B, a.N, a.H, a.hd, a.scale = B, N, n_heads, hd, (1.0 / math.sqrt(hd) if scale is None else float(scale))
    a.dropout_p, a.dropout_seed = dropout_p, dropout_seed
    if dropout_p > 0.0 and STEP_STATE is not None:
        a.seed_off = STEP_STATE.data_ptr()
    check(_launch("attn_fwd_kernel", 4.0 * B * n_heads * N * N * hd,
                  lambda: lib.sfcvit_attention_fwd(ctypes.byref(a), _stream())), "sfcvit_attention_fwd")
    return out, lse


def attention_bwd(qkv, out, lse, dout, n_heads, dropout_p=0.0, dropout_seed=0, colsum=None, scale=None):
    """-> dqkv [, column sums of dqkv over all B * N rows = the in_proj bias gradient: colsum=True -> fp32 [3D] (new
    tensor), or a bf16 [3D] tensor to write into (e.g. a slot of the flat gradient buffer)]."""
    _need(dout, _BF16, "attention dout", 3)
    B, N, D3 = qkv.shape
    D, hd = _attn_dims(D3, n_heads)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, n_heads, N), device=qkv.device, dtype=torch.float32)
    a = _lib.AttnArgs()
    a.qkv, a.out, a.lse, a.dout = qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr()
    a.dqkv, a.delta = dqkv.data_ptr(), delta.data_ptr()
    a.B, a.N, a.H, a.hd, a.scale = B, N, n_heads, hd, (1.0 / math.sqrt(hd) if scale is None else float(scale))
    a.dropout_p, a.dropout_seed = dropout_p, dropout_seed
    if dropout_p > 0.0 and STEP_STATE is not None:
        a.seed_off = STEP_STATE.data_ptr()
    cs = None
    if colsum is not None and colsum is not False:
        cs = torch.empty(D3, device=qkv.device, dtype=torch.float32) if colsum is True else colsum
        if cs.numel() != D3 or not cs.is_contiguous() or cs.dtype not in (torch.float32, _BF16):
            raise ValueError("attention_bwd colsum: contiguous fp32 or bf16 [3D] expected")
        nbytes = lib.sfcvit_attention_colsum_workspace(B, N, n_heads, hd)
        ws = torch.empty(nbytes, device=qkv.device, dtype=torch.uint8)
        a.colsum_part, a.colsum_part_bytes = ws.data_ptr(), nbytes
        a.colsum_out, a.colsum_bf16 = cs.data_ptr(), int(cs.dtype == _BF16)
    with _Deferring([colsum] if cs is not None and colsum is not True else [], [ws] if cs is not None else []):
        check(_launch("attn_bwd", 10.0 * B * n_heads * N * N * hd,
                      lambda: lib.sfcvit_attention_bwd(ctypes.byref(a), _stream())), "sfcvit_attention_bwd")
    return dqkv if cs is None else (dqkv, cs)


# ----------------------------------------------------------------------------
# tokenizer
# ----------------------------------------------------------------------------
class TileDesc:
    """Device copy + host facts of a tile descriptor (sfcvit_tile_descriptors): the tokenizer's pixel table turned out
    to be 16 x 16 pixel tiles (or 256-pixel strips), so the coalesced kernels of csrc/patch_embed_tiled.hip apply."""

    def __init__(self, desc_host, device):
        self.mode, self.ncls = int(desc_host[0]), int(desc_host[1])
        self.cnt = [int(desc_host[6 + c + 1] - desc_host[6 + c]) for c in range(self.ncls)]
        self.dev = torch.from_numpy(desc_host.copy()).to(device)


def tile_descriptor(pix_host, img_w, device):
    """pix_host: numpy int32 [N, P] (the host pixel table) -> TileDesc, or None when the tokens are not tiles / strips."""
    import numpy as np
    N, P = pix_host.shape
    pix_host = np.ascontiguousarray(pix_host, dtype=np.int32)
    cap = 16 + 2 * N + 2 * 8 * 256
    out = np.zeros(cap, dtype=np.int32)
    n = lib.sfcvit_tile_descriptors(ctypes.c_void_p(pix_host.ctypes.data), N, P, int(img_w), ctypes.c_void_p(out.ctypes.data), cap)
    if n < 0:
        check(n, "sfcvit_tile_descriptors")
    return TileDesc(out[:n], device) if n > 0 else None


def _pe_desc(a, desc):
    if desc is not None:
        a.desc, a.desc_ncls = desc.dev.data_ptr(), desc.ncls
        for c, n in enumerate(desc.cnt):
            a.desc_cnt[c] = n


def _pe_args(x, pix, n_tokens, P, D):
    if not x.is_cuda:
        raise _lib.SfcvitError("patch_embed: the HIP path needs a CUDA (ROCm) tensor; there is no CPU fallback")
    _cur_dev(x, "patch_embed x")
    if x.dtype not in (torch.float32, _BF16) or x.dim() != 4 or not x.is_contiguous():
        raise ValueError(f"patch_embed x: expected contiguous [B,C,H,W] fp32/bf16, got {x.dtype} {tuple(x.shape)}")
    _need(pix, torch.int32, "patch_embed pix", 2)
    B, C, H, W = x.shape
    a = _lib.PatchEmbedArgs()
    a.x, a.pix = x.data_ptr(), pix.data_ptr()
    a.B, a.C, a.HW, a.N, a.P, a.D = B, C, H * W, n_tokens, P, D
    a.x_is_bf16 = int(x.dtype == _BF16)
    return a


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def hier_resample_concat(levels):
    """levels: list of [B, N_l, D] bf16 (first = the target length) -> [B, N_0, L * D] bf16: torch's linear interpolation
    (align_corners=False) of every other level to N_0 tokens + the concatenation, one kernel (multi_hilbert.py:33-38)."""
    for t in levels:
        _need(t, _BF16, "hier_resample_concat level", 3)
    B, N0, D = levels[0].shape
    if any(t.shape[0] != B or t.shape[2] != D for t in levels):
        raise ValueError("hier_resample_concat: levels must share batch and width")
    L = len(levels)
    out = torch.empty((B, N0, L * D), device=levels[0].device, dtype=_BF16)
    n = (ctypes.c_int32 * L)(*[t.shape[1] for t in levels])
    check(_launch("hier_resample", 0.0, lambda: lib.sfcvit_hier_resample_concat(_ptr_array(levels), n, L, B, N0, D, _p(out), _stream())),
          "sfcvit_hier_resample_concat")
    return out


def hier_resample_concat_bwd(dout, n_tokens, D):
    """dout [B, N_0, L * D] bf16 -> list of d y_l [B, N_l, D] bf16."""
    _need(dout, _BF16, "hier_resample_concat dout", 3)
    B, N0, LD = dout.shape
    L = len(n_tokens)
    if LD != L * D or n_tokens[0] != N0:
        raise ValueError("hier_resample_concat_bwd: shape mismatch")
    grads = [torch.empty((B, nl, D), device=dout.device, dtype=_BF16) for nl in n_tokens]
    n = (ctypes.c_int32 * L)(*n_tokens)
    check(_launch("hier_resample_bwd", 0.0, lambda: lib.sfcvit_hier_resample_concat_bwd(_p(dout), n, L, B, N0, D, _ptr_array(grads), _stream())),
          "sfcvit_hier_resample_concat_bwd")
    return grads


GATHER_TILES = _os.environ.get("SFCVIT_GATHER_TILES", "1") != "0"     # 0: the per-pixel gather kernel for tile tables too (A/B)


def gather_order(pix_host):
    """Tokens sorted by their lowest pixel offset (numpy int32 [N]): the order in which the gather kernels pair tokens, so
    that horizontally adjacent 16 x 16 tiles -- whose rows share 128-byte lines -- go to one workgroup.  A property of the
    pixel table: the tokenizer computes it once, with the table (a performance hint only, see sfcvit_tokens_gather)."""
    import numpy as np
    return np.argsort(pix_host.min(axis=1), kind="stable").astype(np.int32)


def gather_tokens(x, pix, desc=None, order=None):
    """x [B,C,H,W] fp32/bf16, pix [N,P] int32 (device) -> tokens [B*N, P*C (rounded up to 8)] bf16 in the reference's
    feature order kk * C + c: the A operand of the projection GEMM and of its weight gradient.
    desc: TileDesc of the table or None; order: device int32 [N] from gather_order() or None (tokens paired as numbered).
    16 x 16 tiles of an fp32 image go to sfcvit_tokens_gather_tiles, everything else to sfcvit_tokens_gather."""
    N, P = pix.shape
    a = _pe_args(x, pix, N, P, 8)
    ld = (P * a.C + 7) // 8 * 8
    tokens = torch.empty((a.B * N, ld), device=x.device, dtype=_BF16)
    if order is not None:
        _need(order, torch.int32, "gather order", 1)
    if GATHER_TILES and desc is not None and desc.mode == 1 and P == 256 and a.C <= 4 and x.dtype == torch.float32:
        origin = desc.dev[16 + N:16 + 2 * N]
        check(_launch("tokens_gather", 0.0, lambda: lib.sfcvit_tokens_gather_tiles(_p(x), _p(pix), _p(order), _p(origin), a.B, a.C, x.shape[2],
                                                                                 x.shape[3], N, _p(tokens), ld, _stream())),
              "sfcvit_tokens_gather_tiles")
        return tokens
    check(_launch("tokens_gather", 0.0, lambda: lib.sfcvit_tokens_gather(_p(x), a.x_is_bf16, _p(pix), _p(order), a.B, a.C, a.HW, N, P,
                                                                       _p(tokens), ld, _stream())), "sfcvit_tokens_gather")
    return tokens


def patch_embed_fwd(x, pix, w, bias, desc=None):
    """x [B,C,H,W] fp32/bf16, pix [N,P] int32 (device), w [D, P*C] bf16 -> [B, N, D] bf16.
    desc: TileDesc (tile_descriptor) or None; with it the tiled kernel runs when P = 256 and D % 256 == 0."""
    N, P = pix.shape
    D = w.shape[0]
    _need(w, _BF16, "patch_embed w", 2)
    a = _pe_args(x, pix, N, P, D)
    y = torch.empty((x.shape[0], N, D), device=x.device, dtype=_BF16)
    nbytes = lib.sfcvit_patch_embed_workspace(a.B, a.C, N, P, D, 0)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    a.w, a.y, a.workspace, a.workspace_bytes = w.data_ptr(), y.data_ptr(), ws.data_ptr(), nbytes
    a.bias = _need(bias, _BF16, "patch_embed bias", 1).data_ptr() if bias is not None else None
    _pe_desc(a, desc)
    check(_launch("pe_fwd_kernel", 2.0 * a.B * N * P * a.C * D,
                  lambda: lib.sfcvit_patch_embed_fwd(ctypes.byref(a), _stream())), "sfcvit_patch_embed_fwd")
    return y


def patch_embed_bwd(x, pix, dy, D, want_bias=True, desc=None):
    """-> dW fp32 [D, P*C], dbias fp32 [D]."""
    N, P = pix.shape
    _need(dy, _BF16, "patch_embed dy", 3)
    a = _pe_args(x, pix, N, P, D)
    dw = torch.empty((D, P * a.C), device=x.device, dtype=torch.float32)
    db = torch.empty(D, device=x.device, dtype=torch.float32) if want_bias else None
    nbytes = lib.sfcvit_patch_embed_workspace(a.B, a.C, N, P, D, 1)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    a.y, a.dw, a.dbias = dy.data_ptr(), dw.data_ptr(), (db.data_ptr() if want_bias else None)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    _pe_desc(a, desc)
    check(_launch("pe_bwd_kernel", 2.0 * a.B * N * P * a.C * D,
                  lambda: lib.sfcvit_patch_embed_bwd(ctypes.byref(a), _stream())), "sfcvit_patch_embed_bwd")
    return dw, db


def hier_tokenizer_supported(n_levels, D, C, pixels_per_token):
    """Is (levels, level width, channels, pixels per token of each level) inside the fused hierarchical kernel's
    envelope (sfcvit_hier_tokenizer_supported)?  Host arithmetic only."""
    if not 1 <= n_levels <= 4 or len(pixels_per_token) != n_levels:
        return False
    P = (ctypes.c_int32 * 4)(*pixels_per_token)
    return bool(lib.sfcvit_hier_tokenizer_supported(n_levels, D, C, P))


def hier_tokenizer_fwd(x, pix_list, w_list, b_list, wf, bf):
    """Fused hierarchical tokenizer (csrc/hier_tokenizer.hip): x [B,C,H,W] fp32/bf16; per level pix [N,P_l] int32,
    w [D,P_l*C] bf16, b [D] bf16 or None; wf [L*D,L*D], bf [L*D] or None -> (y [B,N,L*D] bf16, h [B,N,L*D] bf16).
    wf = None: gather + level projections + concatenation only -> (None, h); the caller applies the fusion Linear."""
    L = len(pix_list)
    if not x.is_cuda:
        raise _lib.SfcvitError("hier_tokenizer: the HIP path needs a CUDA (ROCm) tensor; there is no CPU fallback")
    _cur_dev(x, "hier_tokenizer x")
    if x.dtype not in (torch.float32, _BF16) or x.dim() != 4 or not x.is_contiguous():
        raise ValueError(f"hier_tokenizer x: expected contiguous [B,C,H,W] fp32/bf16, got {x.dtype} {tuple(x.shape)}")
    B, C, H, W = x.shape
    N = pix_list[0].shape[0]
    D = w_list[0].shape[0]
    E = L * D
    a = _lib.HierArgs()
    a.x, a.x_is_bf16 = x.data_ptr(), int(x.dtype == _BF16)
    flops = 2.0 * B * N * E * E if wf is not None else 0.0
    for l in range(L):
        pix = _need(pix_list[l], torch.int32, "hier_tokenizer pix", 2)
        if pix.shape[0] != N:
            raise ValueError("hier_tokenizer: every level must have the same token count")
        w = _need(w_list[l], _BF16, "hier_tokenizer w", 2)
        if tuple(w.shape) != (D, pix.shape[1] * C):
            raise ValueError(f"hier_tokenizer: level {l} weight {tuple(w.shape)}, expected {(D, pix.shape[1] * C)}")
        a.pix[l], a.w[l], a.P[l] = pix.data_ptr(), w.data_ptr(), pix.shape[1]
        a.b[l] = _need(b_list[l], _BF16, "hier_tokenizer b", 1).data_ptr() if b_list[l] is not None else None
        flops += 2.0 * B * N * pix.shape[1] * C * D
    y = None
    if wf is not None:
        _need(wf, _BF16, "hier_tokenizer wf", 2)
        if tuple(wf.shape) != (E, E):
            raise ValueError(f"hier_tokenizer: fusion weight {tuple(wf.shape)}, expected {(E, E)}")
        a.wf = wf.data_ptr()
        a.bf = _need(bf, _BF16, "hier_tokenizer bf", 1).data_ptr() if bf is not None else None
        y = torch.empty((B, N, E), device=x.device, dtype=_BF16)
        a.y = y.data_ptr()
    h = torch.empty((B, N, E), device=x.device, dtype=_BF16)
    a.h = h.data_ptr()
    a.B, a.C, a.HW, a.N, a.L, a.D = B, C, H * W, N, L, D
    check(_launch("hier_fwd_kernel", flops, lambda: lib.sfcvit_hier_tokenizer_fwd(ctypes.byref(a), _stream())),
          "sfcvit_hier_tokenizer_fwd")
    return y, h


# ----------------------------------------------------------------------------
# elementwise / loss / optimizer
# ----------------------------------------------------------------------------
def gelu_fwd(x):
    _need(x, _BF16, "gelu x")
    y = torch.empty_like(x)
    check(lib.sfcvit_gelu_fwd(_p(x), _p(y), x.numel(), _stream()), "sfcvit_gelu_fwd")
    return y


def gelu_bwd(dy, x):
    _need(x, _BF16, "gelu x")
    _need(dy, _BF16, "gelu dy")
    dx = torch.empty_like(x)
    check(lib.sfcvit_gelu_bwd(_p(dy), _p(x), _p(dx), x.numel(), _stream()), "sfcvit_gelu_bwd")
    return dx


def gelu_drop_fwd(x, p, seed):
    _need(x, _BF16, "gelu_drop x", 2)
    y = torch.empty_like(x)
    check(lib.sfcvit_gelu_drop_fwd(_p(x), _p(y), x.shape[0], x.shape[1], p, seed, _seed_off(), _stream()), "sfcvit_gelu_drop_fwd")
    return y


def gelu_drop_bwd(dy, x, p, seed):
    _need(x, _BF16, "gelu_drop x", 2)
    _need(dy, _BF16, "gelu_drop dy", 2)
    dx = torch.empty_like(x)
    check(lib.sfcvit_gelu_drop_bwd(_p(dy), _p(x), _p(dx), x.shape[0], x.shape[1], p, seed, _seed_off(), _stream()),
          "sfcvit_gelu_drop_bwd")
    return dx


def dropout_mask(rows, cols, p, seed, device="cuda"):
    """The keep mask as bf16 {0, 1/(1-p)} [rows, cols] (tests / debugging)."""
    out = torch.empty((rows, cols), device=device, dtype=_BF16)
    check(lib.sfcvit_dropout_mask(_p(out), rows, cols, p, seed, _stream()), "sfcvit_dropout_mask")
    return out


def soft_ce(logits, targets, n_classes, gscale):
    """logits bf16 [B, ld] (first n_classes columns are classes), targets fp32 [B, C].
    -> loss_rows fp32 [B], dlogits bf16 [B, ld] (already multiplied by gscale)."""
    _need(logits, _BF16, "soft_ce logits", 2)
    _need(targets, torch.float32, "soft_ce targets", 2)
    B, ld = logits.shape
    loss_rows = torch.empty(B, device=logits.device, dtype=torch.float32)
    dlogits = torch.empty_like(logits)
    check(lib.sfcvit_soft_ce(_p(logits), _p(targets), _p(loss_rows), _p(dlogits), B, n_classes, ld, gscale, _stream()),
          "sfcvit_soft_ce")
    return loss_rows, dlogits


def sumsq_accum(g, out):
    is_f32 = g.dtype == torch.float32
    if not is_f32:
        _need(g, _BF16, "sumsq g")
    ws = torch.empty(4096, device=g.device, dtype=torch.uint8)           # SFCVIT_SUMSQ_WORKSPACE_BYTES
    check(lib.sfcvit_sumsq_accum(_p(g), g.numel(), int(is_f32), _p(out), _p(ws), _stream()), "sfcvit_sumsq_accum")


def adamw_step(param, master, grad, m, v, sumsq, *, lr, beta1, beta2, eps, weight_decay, max_norm, step,
               grad_scale=1.0, dev_state=None):
    a = _lib.AdamWArgs()
    a.param, a.master, a.grad, a.m, a.v = (param.data_ptr(), master.data_ptr(), grad.data_ptr(),
                                           m.data_ptr(), v.data_ptr())
    a.sumsq = sumsq.data_ptr() if sumsq is not None else None
    a.n = param.numel()
    a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.max_norm, a.step = lr, beta1, beta2, eps, weight_decay, max_norm, step
    a.grad_scale = grad_scale
    if dev_state is not None:
        a.dev_state = dev_state.data_ptr()
    check(lib.sfcvit_adamw_step(ctypes.byref(a), _stream()), "sfcvit_adamw_step")
