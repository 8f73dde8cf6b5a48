"""ctypes binding of libsfcvit_hip.so (the C ABI declared in include/sfcvit.h).

The library is the product: there is no CPU or PyTorch fallback behind it.  If it
is missing or does not load, importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFCVIT_LIB", os.path.join(_HERE, "libsfcvit_hip.so"))   # override: A/B builds of the kernels

c_void_p, c_int, c_int32, c_int64, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int32,
                                              ctypes.c_int64, ctypes.c_float)


class PatchEmbedArgs(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("pix", c_void_p), ("w", c_void_p), ("bias", c_void_p),
                ("y", c_void_p), ("dw", c_void_p), ("dbias", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_int64),
                ("B", c_int32), ("C", c_int32), ("HW", c_int32), ("N", c_int32), ("P", c_int32),
                ("D", c_int32), ("x_is_bf16", c_int32),
                ("desc", c_void_p), ("desc_ncls", c_int32), ("desc_cnt", c_int32 * 8)]


class HierArgs(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("pix", c_void_p * 4), ("w", c_void_p * 4), ("b", c_void_p * 4),
                ("wf", c_void_p), ("bf", c_void_p), ("h", c_void_p), ("y", c_void_p),
                ("P", c_int32 * 4), ("B", c_int32), ("C", c_int32), ("HW", c_int32), ("N", c_int32), ("L", c_int32),
                ("D", c_int32), ("x_is_bf16", c_int32)]


class GemmArgs(ctypes.Structure):
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("c", c_void_p), ("bias", c_void_p),
                ("residual", c_void_p), ("aux_in", c_void_p), ("aux_out", c_void_p),
                ("M", c_int32), ("N", c_int32), ("K", c_int32),
                ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32), ("ldr", c_int32), ("ldaux", c_int32),
                ("a_kmajor", c_int32), ("b_kmajor", c_int32), ("act", c_int32), ("dact", c_int32),
                ("c_is_f32", c_int32), ("splitk", c_int32), ("workspace", c_void_p),
                ("workspace_bytes", c_int64), ("force_generic", c_int32),
                ("dropout_p", c_float), ("dropout_seed", ctypes.c_uint32), ("dact_scale", c_float),
                ("row_offset", c_int32), ("colsum_out", c_void_p), ("colsum_bf16", c_int32), ("seed_off", c_void_p),
                ("actmask", c_void_p), ("ld_actmask", c_int32)]


class AttnArgs(ctypes.Structure):
    _fields_ = [("qkv", c_void_p), ("out", c_void_p), ("lse", c_void_p), ("dout", c_void_p),
                ("dqkv", c_void_p), ("delta", c_void_p),
                ("B", c_int32), ("N", c_int32), ("H", c_int32), ("hd", c_int32), ("scale", c_float),
                ("dropout_p", c_float), ("dropout_seed", ctypes.c_uint32), ("seed_off", c_void_p),
                ("colsum_part", c_void_p), ("colsum_part_bytes", c_int64), ("colsum_out", c_void_p), ("colsum_bf16", c_int32)]


class AdamWArgs(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("master", c_void_p), ("grad", c_void_p), ("m", c_void_p),
                ("v", c_void_p), ("sumsq", c_void_p), ("n", c_int64),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("weight_decay", c_float), ("max_norm", c_float), ("grad_scale", c_float),
                ("step", c_int32), ("dev_state", c_void_p)]


# name -> (restype, argtypes); every symbol include/sfcvit.h declares.
SIGNATURES = {
    "sfcvit_abi_version": (c_int, []),
    "sfcvit_last_error": (ctypes.c_char_p, []),
    "sfcvit_device_count": (c_int, []),
    "sfcvit_curve_table": (c_int, [c_int, c_int, c_void_p]),
    "sfcvit_curve_table_rc": (c_int, [c_int, c_int, c_void_p]),
    "sfcvit_pixel_table": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "sfcvit_tile_descriptors": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int]),
    "sfcvit_patch_embed_fwd": (c_int, [ctypes.POINTER(PatchEmbedArgs), c_void_p]),
    "sfcvit_hier_tokenizer_supported": (c_int, [c_int, c_int, c_int, c_void_p]),
    "sfcvit_hier_tokenizer_fwd": (c_int, [ctypes.POINTER(HierArgs), c_void_p]),
    "sfcvit_patch_embed_bwd": (c_int, [ctypes.POINTER(PatchEmbedArgs), c_void_p]),
    "sfcvit_gemm": (c_int, [ctypes.POINTER(GemmArgs), c_void_p]),
    "sfcvit_gemm_workspace": (c_int64, [c_int, c_int, c_int]),
    "sfcvit_patch_embed_workspace": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "sfcvit_transpose": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sfcvit_transpose_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sfcvit_hier_resample_concat": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sfcvit_hier_resample_concat_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sfcvit_tokens_gather_tiles": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sfcvit_tokens_gather": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sfcvit_reduce_defer": (c_int, [c_int]),
    "sfcvit_reduce_pending": (c_int, []),
    "sfcvit_reduce_flush": (c_int, [c_void_p]),
    "sfcvit_reduce_discard": (c_int, []),
    "sfcvit_gemm_colsum_workspace": (c_int64, [c_int, c_int]),
    "sfcvit_last_gemm_kernel": (c_int, [ctypes.c_char_p, c_int]),
    "sfcvit_last_attn_kernel": (c_int, [ctypes.c_char_p, c_int]),
    "sfcvit_last_rowwise_kernel": (c_int, [ctypes.c_char_p, c_int]),
    "sfcvit_colsum_workspace": (c_int64, [c_int, c_int]),
    "sfcvit_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "sfcvit_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_float, c_void_p]),
    "sfcvit_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sfcvit_layernorm_bwd_drop": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_float, ctypes.c_uint32, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sfcvit_layernorm_bwd_ws": (c_int64, [c_int, c_int]),
    "sfcvit_gelu_drop_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, ctypes.c_uint32, c_void_p, c_void_p]),
    "sfcvit_gelu_drop_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, ctypes.c_uint32, c_void_p, c_void_p]),
    "sfcvit_step_advance": (c_int, [c_void_p, c_float, c_float, ctypes.c_uint32, c_void_p]),
    "sfcvit_dropout_mask": (c_int, [c_void_p, c_int64, c_int, c_float, ctypes.c_uint32, c_void_p]),
    "sfcvit_attention_fwd": (c_int, [ctypes.POINTER(AttnArgs), c_void_p]),
    "sfcvit_attention_bwd": (c_int, [ctypes.POINTER(AttnArgs), c_void_p]),
    "sfcvit_attention_colsum_workspace": (c_int64, [c_int, c_int, c_int, c_int]),
    "sfcvit_gelu_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "sfcvit_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sfcvit_soft_ce": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_float, c_void_p]),
    "sfcvit_sumsq_accum": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "sfcvit_adamw_step": (c_int, [ctypes.POINTER(AdamWArgs), c_void_p]),
}


class SfcvitError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or `make -C "
            f"{os.path.join(os.path.dirname(_HERE), 'csrc')}`).  There is no fallback path.")
    try:
        # torch must own the HIP runtime of the process: loaded first, its libamdhip64.so
        # (soname libamdhip64.so.7) satisfies this library's dependency as well, so kernels
        # run on torch's streams and memory.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.sfcvit_abi_version() != 1:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.sfcvit_abi_version()} != 1")
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        msg = lib.sfcvit_last_error().decode(errors="replace")
        raise SfcvitError(f"{what} failed (status {rc}): {msg}")
