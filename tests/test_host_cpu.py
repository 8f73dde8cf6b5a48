"""Host-side logic of the product, no GPU: native curve tables against the oracle and
the golden fixtures, the C-ABI surface, the nn.Module / state_dict surface."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import curves as ocurves
from oracle.cases import MODEL_CASES, CURVE_KINDS, CURVE_SMALL_N, CURVE_SHA_N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sfc():
    from sfcvit.curves import space_filling_curves as m
    return m


@pytest.mark.parametrize("kind", CURVE_KINDS)
@pytest.mark.parametrize("n", CURVE_SMALL_N)
def test_native_tables_match_golden(kind, n, sfc, golden_dir):
    small = np.load(os.path.join(golden_dir, "curves_small.npz"))
    assert np.array_equal(sfc.curve_table(kind, n), small[f"{kind}_{n}"])


@pytest.mark.parametrize("kind", ("hilbert", "z"))
@pytest.mark.parametrize("n", CURVE_SHA_N)
def test_native_tables_full_size(kind, n, sfc, golden_dir):
    with open(os.path.join(golden_dir, "curves_sha.json")) as f:
        sha = json.load(f)
    assert hashlib.sha256(sfc.curve_table(kind, n).tobytes()).hexdigest() == sha[f"{kind}_{n}"]


@pytest.mark.parametrize("kind", CURVE_KINDS)
@pytest.mark.parametrize("n", [1, 3, 5, 7, 9, 27, 33, 63, 64, 65, 100, 128, 243, 257, 512])
def test_native_tables_match_oracle(kind, n, sfc):
    # integer closed forms (product) vs float64 recursion (oracle), ragged and maximum sizes
    assert np.array_equal(sfc.curve_table(kind, n).astype(np.int64), ocurves.flat_table(kind, n))
    rc = sfc.curve_table_rc(kind, n)
    assert np.array_equal(rc, ocurves.embed_and_prune_sfc(kind, n, n))


def test_rectangular_embed_and_prune(sfc):
    for fn, kind in ((sfc.hilbert_curve, "hilbert"), (sfc.z_curve, "z"), (sfc.peano_curve, "peano")):
        for w, h in ((3, 2), (5, 12), (16, 4), (10, 10)):
            got = sfc.embed_and_prune_sfc(fn, w, h)
            ref = [tuple(int(v) for v in p) for p in ocurves.embed_and_prune_sfc(kind, w, h)]
            assert got == ref
    with pytest.raises(ValueError):
        sfc.embed_and_prune_sfc(lambda o, s: [], 4, 4)      # unknown curve, as the reference's grid_size


def test_curve_functions_return_cell_centres(sfc):
    pts = sfc.hilbert_curve(2, size=4)
    assert len(pts) == 16 and pts[0] == (0.5, 0.5)
    cells = [(int(np.floor(x)), int(np.floor(y))) for x, y in pts]
    assert cells == sfc.embed_and_prune_sfc(sfc.hilbert_curve, 4, 4)


def test_pixel_table_matches_reference_semantics():
    from sfcvit.tokenizers.embeddings import _pixel_table
    from oracle import vit_oracle
    # p = 2, g = 4 on an 8x8 image: compare with the oracle's rearrange-based tokenisation
    img, p, g = 8, 2, 4
    flat = ocurves.flat_table("hilbert", img // p)
    pix = _pixel_table(flat, img, p, g)
    x = torch.arange(img * img, dtype=torch.float32).reshape(1, 1, img, img)
    tok = vit_oracle.tokens_sfc(x, torch.from_numpy(flat), p, g)[0]          # [N, g*p*p] (C = 1)
    assert np.array_equal(pix, tok.numpy().astype(np.int32))


def test_library_exports_every_declared_symbol():
    from sfcvit import _lib
    header = open(os.path.join(ROOT, "include", "sfcvit.h")).read()
    declared = set(re.findall(r"\b(sfcvit_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert _lib.lib.sfcvit_abi_version() == 1


def test_argument_checks_run_without_a_gpu():
    from sfcvit import _lib
    args = _lib.GemmArgs()
    assert _lib.lib.sfcvit_gemm(ctypes.byref(args), None) == 1          # SFCVIT_EINVAL, nothing launched
    assert b"null" in _lib.lib.sfcvit_last_error()
    out = np.zeros(4, dtype=np.int32)
    assert _lib.lib.sfcvit_curve_table(99, 2, out.ctypes.data_as(ctypes.c_void_p)) == 1
    assert _lib.lib.sfcvit_curve_table(0, 0, out.ctypes.data_as(ctypes.c_void_p)) == 1


def build_model(cfg):
    from sfcvit.tokenizers import HilbertEmbedding1D, MortonEmbedding1D, RasterScan1DEmbedding, SFCEmbedding1D
    from sfcvit.models import VisionTransformer, VisionTransformer1D
    from sfcvit.curves import hilbert_curve, z_curve
    if cfg.tokenizer == "hilbert1d":
        pe = HilbertEmbedding1D(cfg.img_size, cfg.patch_size, cfg.in_channels, cfg.embed_dim)
    elif cfg.tokenizer == "morton1d":
        pe = MortonEmbedding1D(cfg.img_size, cfg.patch_size, cfg.in_channels, cfg.embed_dim)
    elif cfg.tokenizer == "raster1d":
        pe = RasterScan1DEmbedding(cfg.img_size, cfg.patch_size, cfg.in_channels, cfg.embed_dim)
    else:
        pe = SFCEmbedding1D(cfg.img_size, cfg.pre_patch_size, cfg.patch_size, cfg.in_channels, cfg.embed_dim,
                            {"hilbert": hilbert_curve, "z": z_curve}[cfg.curve])
    cls = VisionTransformer1D if cfg.variant == "1d" else VisionTransformer
    return cls(pe, depth=cfg.depth, n_heads=cfg.n_heads, mlp_dim=cfg.mlp_dim, num_classes=cfg.num_classes)


@pytest.mark.parametrize("name", sorted(MODEL_CASES))
def test_state_dict_surface_matches_reference(name, golden_dir):
    with open(os.path.join(golden_dir, "state_manifest.json")) as f:
        manifest = json.load(f)[name]
    cfg, _ = MODEL_CASES[name]
    model = build_model(cfg)
    sd = model.state_dict()
    assert sorted(sd) == sorted(manifest)                   # same keys (fixture is stored sorted)
    for k, v in sd.items():
        assert list(v.shape) == manifest[k][0], k
        assert str(v.dtype).replace("torch.", "") == manifest[k][1], k
    # the tokenizer is the same module under both names (vit.py:221)
    assert model.encoder.to_patch_embedding is model.patch_embed
    assert len(list(model.parameters())) == len({id(p) for p in model.parameters()})
    # registered index buffer == reference table (bit-exact)
    from oracle import vit_oracle
    buf = vit_oracle.curve_buffer(cfg)
    if buf is not None:
        assert torch.equal(sd["patch_embed." + cfg.table_key], buf)


def test_same_seed_same_init_as_torch_containers():
    # parameters are created by the same torch constructors in the same order as the reference
    cfg, _ = MODEL_CASES["hilbert32_1d"]
    torch.manual_seed(42)
    a = build_model(cfg).state_dict()
    torch.manual_seed(42)
    b = build_model(cfg).state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(a["encoder.transformer.layers.0.linear1.weight"], a["encoder.transformer.layers.1.linear1.weight"])


def test_cpu_tensors_are_rejected_loudly():
    from sfcvit._lib import SfcvitError
    cfg, batch = MODEL_CASES["hilbert32_1d"]
    model = build_model(cfg).eval()
    with pytest.raises(SfcvitError, match="no CPU fallback"):
        model(torch.zeros(batch, 3, 32, 32))


def test_training_mode_on_cpu_is_rejected_too():
    from sfcvit._lib import SfcvitError
    cfg, batch = MODEL_CASES["hilbert32_1d"]
    model = build_model(cfg).train()            # dropout 0.1 / 0.5 as in the reference: fused in the HIP kernels
    with pytest.raises(SfcvitError, match="no CPU fallback"):
        model(torch.zeros(batch, 3, 32, 32))


def test_reference_import_paths_resolve_to_the_hip_modules():
    """A script written against the reference's package layout (`from src.… import …`) runs unchanged with
    space-filling-curves-for-vision-transformers_amd/ on sys.path."""
    import importlib
    import sfcvit.models.vit as vit
    import sfcvit.tokenizers as tok
    pairs = [("src.models.vit", "VisionTransformer1D", vit.VisionTransformer1D),
             ("src.models.vit", "VisionTransformer", vit.VisionTransformer),
             ("src.tokenizers._1D.hilbert_embedding1D", "HilbertEmbedding1D", tok.HilbertEmbedding1D),
             ("src.tokenizers._1D.morton_embedding1D", "MortonEmbedding1D", tok.MortonEmbedding1D),
             ("src.tokenizers._1D.zigzag_embedding1D", "RasterScan1DEmbedding", tok.RasterScan1DEmbedding),
             ("src.tokenizers.multiscale.multi_hilbert", "HierarchicalHilbertEmbedding", tok.HierarchicalHilbertEmbedding),
             ("src.tokenizers.multiscale.multi_hilbert", "SFCEmbedding1D", tok.SFCEmbedding1D),
             ("src.tokenizers.multiscale.multi_morton", "HierarchicalMortonEmbedding", tok.HierarchicalMortonEmbedding)]
    for mod, name, obj in pairs:
        assert getattr(importlib.import_module(mod), name) is obj, (mod, name)
    for mod, names in [("src.curves.space_filling_curves", ["hilbert_curve", "z_curve", "embed_and_prune_sfc"]),
                       ("src.training.train", ["train_with_mixup_or_cutmix", "mixup_data", "cutmix_data", "rand_bbox",
                                               "mixup_criterion", "evaluate"])]:
        m = importlib.import_module(mod)
        for n in names:
            assert callable(getattr(m, n)), (mod, n)


def test_reference_default_model_has_the_reference_parameter_count(golden_dir):
    """main.py:269-282: HierarchicalMortonEmbedding(32, 3, [16, 4, 1], 256) + 8 layers."""
    from sfcvit.models.vit import VisionTransformer1D
    from sfcvit.tokenizers import HierarchicalMortonEmbedding
    pe = HierarchicalMortonEmbedding(32, 3, [16, 4, 1], 256)
    assert (pe.n_patches, pe.embed_dim) == (64, 768)
    model = VisionTransformer1D(pe, depth=8, n_heads=4, mlp_dim=512, num_classes=10)
    assert sum(p.numel() for p in model.parameters()) == 34_773_834
    with open(os.path.join(golden_dir, "hierarchical.json")) as f:
        gold = json.load(f)["hier_morton32"]
    small = HierarchicalMortonEmbedding(32, 3, [16, 4, 1], 64)
    assert {k: list(v.shape) for k, v in small.state_dict().items()} == gold["keys"]


def test_mixup_cutmix_helpers_follow_the_reference_contract():
    """src/training/train.py:9-87: lam ~ Beta(alpha, alpha); CutMix pastes one box from the permuted batch and
    corrects lam to the pasted area."""
    from sfcvit.training.loops import cutmix_data, mixup_data, rand_bbox
    torch.manual_seed(0)
    np.random.seed(0)
    x = torch.randn(8, 3, 16, 16)
    y = torch.arange(8)
    mx, ya, yb, lam = mixup_data(x, y, alpha=1.0)
    assert mx.shape == x.shape and 0.0 <= lam <= 1.0 and torch.equal(ya, y) and sorted(yb.tolist()) == list(range(8))
    perm = yb
    assert torch.allclose(mx, lam * x + (1 - lam) * x[perm], atol=1e-6)
    x1, y1 = rand_bbox(16, 16, 0.75)[0:2]
    cx, ya, yb, lam = cutmix_data(x.clone(), y, alpha=1.0)
    changed = (cx != x).any(dim=1).any(dim=0)               # [H, W] mask of pasted pixels
    assert abs((1.0 - changed.float().mean().item()) - lam) < 1e-6 or changed.sum() == 0


def test_spiral_and_2d_hilbert_native_tables_match_reference_fixture(sfc, golden_dir):
    from oracle.cases import SPIRAL_N, HILBERT_T_N
    gold = np.load(os.path.join(golden_dir, "curves_extra.npz"))
    for n in SPIRAL_N:
        assert np.array_equal(sfc.curve_table(sfc.spiral_curve, n), gold[f"spiral_{n}"]), n
    for n in HILBERT_T_N:
        assert np.array_equal(sfc.curve_table(sfc.hilbert_t_curve, n), gold[f"hilbert_t_{n}"]), n
    for n in (224, 384, 77):
        assert np.array_equal(sfc.curve_table(sfc.spiral_curve, n), ocurves.flat_table("spiral", n)), n
    with pytest.raises(Exception, match="power of two"):
        sfc.curve_table(sfc.hilbert_t_curve, 14)


def _tok_cases():
    from oracle.cases import TOKENIZER_CASES
    return sorted(TOKENIZER_CASES)


@pytest.mark.parametrize("name", _tok_cases())
def test_remaining_tokenizers_have_the_reference_surface(name, golden_dir):
    """Same import path (through src.*), constructor, state_dict keys/shapes and n_patches / embed_dim attributes."""
    import importlib
    from oracle.cases import TOKENIZER_CASES
    modname, clsname, args, kind, _ = TOKENIZER_CASES[name]
    with open(os.path.join(golden_dir, "tokenizers.json")) as f:
        gold = json.load(f)[name]
    mod = getattr(importlib.import_module(modname), clsname)(*args)
    assert {k: list(v.shape) for k, v in mod.state_dict().items()} == gold["keys"]
    assert int(getattr(mod, "n_patches", getattr(mod, "n_final_patches", -1))) == gold["n_patches"] or gold["n_patches"] == -1
    if hasattr(mod, "embed_dim"):
        assert mod.embed_dim == gold["embed_dim"]
    from oracle import vit_oracle
    sd = vit_oracle.tokenizer_case_state(args, kind)
    for k, v in mod.state_dict().items():
        if not v.is_floating_point():
            assert torch.equal(v, sd[k]), k                 # index buffers identical to the oracle's


def test_warmup_cosine_scheduler_values():
    """src/training/scheduler.py:33-51: lr(step) for warm-up 4, total 12, base 1e-3, floor 1e-6."""
    import math
    from src.training.scheduler import WarmupCosineScheduler

    class Opt:
        param_groups = [{"lr": 1e-3}, {"lr": 5.0}]
    sched = WarmupCosineScheduler(Opt, warmup_steps=4, total_steps=12, min_lr=1e-6)
    got = [sched.step() for _ in range(15)]
    want = [1e-3 * s / 4 for s in range(4)] + \
           [1e-6 + 0.5 * (1e-3 - 1e-6) * (1 + math.cos(math.pi * min(1.0, (s - 4) / 8))) for s in range(4, 15)]
    assert got == pytest.approx(want, rel=1e-12)
    assert Opt.param_groups[0]["lr"] == Opt.param_groups[1]["lr"] == got[-1] == pytest.approx(1e-6)


def test_altvit_classes_have_the_reference_surface(golden_dir):
    from oracle.cases import ALTVIT_CASES
    import src.models.altvit as alt
    with open(os.path.join(golden_dir, "altvit.json")) as f:
        gold = json.load(f)
    for name, (clsname, kw, _) in ALTVIT_CASES.items():
        mod = getattr(alt, clsname)(**kw)
        assert {k: list(v.shape) for k, v in mod.state_dict().items()} == gold[name]["keys"], name
        assert abs(float(mod.pos_embedding.double().norm()) - gold[name]["pos_embedding_l2"]) < 1e-4 * gold[name]["pos_embedding_l2"]
    with pytest.raises(AssertionError, match="power of 2"):
        alt.HilbertViT(image_size=48, patch_size=4, num_classes=10, dim=64, depth=1, heads=1, mlp_dim=64)


@pytest.mark.parametrize("curve,img,expect", [("hilbert", 224, (1, 4)), ("hilbert", 384, (1, 4)), ("hilbert", 32, (1, 3)),
                                              ("z", 224, (1, 1)), ("z", 384, (1, 1)), ("raster", 224, (2, 1)), ("raster", 32, (2, 1))])
def test_tile_descriptors_reconstruct_the_pixel_table(curve, img, expect):
    """sfcvit_tile_descriptors (host): SURVEY App. A.5 as code -- at the BASELINE sizes every 256-pixel token of the
    Hilbert / Z order is one 16 x 16 tile with one of <= 4 (Hilbert) / exactly 1 (Z) intra-tile pixel orders, a raster
    token is a 256-pixel strip; the descriptor (origin per token, permutation per class) must reproduce the pixel
    table the generic kernel uses, entry by entry (index work: exact)."""
    import ctypes
    from sfcvit._lib import lib
    from sfcvit.tokenizers.embeddings import _pixel_table
    from sfcvit.curves import curve_table, hilbert_curve, z_curve
    flat = np.arange(img * img, dtype=np.int32) if curve == "raster" else curve_table({"hilbert": hilbert_curve, "z": z_curve}[curve], img)
    pix = _pixel_table(flat, img, 1, 256)
    N = pix.shape[0]
    cap = 16 + 2 * N + 2 * 8 * 256
    desc = np.zeros(cap, dtype=np.int32)
    n = lib.sfcvit_tile_descriptors(ctypes.c_void_p(pix.ctypes.data), N, 256, img, ctypes.c_void_p(desc.ctypes.data), cap)
    assert n == 16 + 2 * N + 2 * expect[1] * 256
    mode, ncls = int(desc[0]), int(desc[1])
    assert (mode, ncls) == expect and desc[4] == N
    starts = desc[6:6 + ncls + 1]
    assert starts[0] == 0 and starts[-1] == N
    toks, origin = desc[16:16 + N], desc[16 + N:16 + 2 * N]
    perm = desc[16 + 2 * N:16 + 2 * N + ncls * 256].reshape(ncls, 256)
    inv = desc[16 + 2 * N + ncls * 256:n].reshape(ncls, 256)
    for c in range(ncls):
        assert np.array_equal(inv[c][perm[c]], np.arange(256))
    assert sorted(toks.tolist()) == list(range(N))
    j = np.arange(256)
    local = (j // 16) * img + j % 16 if mode == 1 else j
    for c in range(ncls):
        assert sorted(perm[c].tolist()) == list(range(256))
        for t in toks[starts[c]:starts[c + 1]]:
            assert np.array_equal(pix[t][perm[c]], origin[t] + local)


def test_tile_descriptors_decline_other_tokenizers():
    import ctypes
    from sfcvit._lib import lib
    from sfcvit.tokenizers.embeddings import _pixel_table
    from sfcvit.curves import curve_table, hilbert_curve
    desc = np.zeros(16 + 2 * 4096 + 4096, dtype=np.int32)
    # 64 pixels per token: not 256
    pix = _pixel_table(curve_table(hilbert_curve, 32), 32, 1, 64)
    assert lib.sfcvit_tile_descriptors(ctypes.c_void_p(pix.ctypes.data), pix.shape[0], 64, 32, ctypes.c_void_p(desc.ctypes.data), desc.size) == 0
    # 256 pixels per token that are not one tile: 2 x 2 pre-patches grouped by 64 along the curve on a 64-px image
    pix = _pixel_table(curve_table(hilbert_curve, 32), 64, 2, 64)
    got = lib.sfcvit_tile_descriptors(ctypes.c_void_p(pix.ctypes.data), pix.shape[0], 256, 64, ctypes.c_void_p(desc.ctypes.data), desc.size)
    assert got >= 0          # a 16 x 16 tile in a different pixel order is still a tile; anything else is declined (0)


def test_hier_tokenizer_envelope_and_padded_head_dims():
    """Host-side decisions that pick kernels: the fused hierarchical tokenizer's envelope (sfcvit_hier_tokenizer_supported)
    and the head dim a model head dim runs on (ops.padded_head_dim)."""
    from sfcvit import ops
    ok = ops.hier_tokenizer_supported
    assert ok(3, 256, 3, [16, 16, 16])                 # main.py:269-274: [16, 4, 1] at 32 x 32 -> 16 pixels per token, 3 x 256
    assert ok(4, 64, 3, [64, 64, 64, 64])              # 4 levels of 192 features
    assert ok(1, 256, 3, [256])
    assert not ok(3, 64, 3, [16, 16, 16])              # L * D = 192 is not a multiple of 256
    assert not ok(2, 96, 3, [16, 16])                  # D % 64
    assert not ok(2, 128, 3, [4, 4])                   # 12 features: not a multiple of 8
    assert not ok(5, 256, 3, [16] * 5)                 # more than 4 levels
    assert not ok(4, 256, 3, [256] * 4)                # 64 rows of (1024 + 4 x 768) bf16 exceed the LDS
    assert not ok(2, 128, 3, [16])                     # one entry per level
    assert [ops.padded_head_dim(h) for h in (16, 32, 48, 64, 96, 128, 160, 192, 200, 256)] == [64, 64, 64, 64, 128, 128, 192, 192, 256, 256]
    assert ops.padded_head_dim(257) is None


def test_hierarchical_fusable_requires_levels_that_cover_the_image():
    """`_fusable` must send a level whose tokens do not cover the image exactly (N * P_l != H * W) to the
    composed path instead of letting sfcvit_hier_tokenizer_fwd reject it.  Host arithmetic only (a stub stands in for
    the CUDA input)."""
    from sfcvit.tokenizers.multiscale import HierarchicalMortonEmbedding

    class X:
        is_cuda, shape = True, (2, 3, 32, 32)

    tok = HierarchicalMortonEmbedding(32, 3, [16, 4, 1], 256)
    assert tok._fusable(X())
    tok._fuse_key = None                                           # the decision is cached per channel count
    tok.levels[2].input_dim *= 2                                   # a level with twice the pixels per token: 64 * 32 != 32 * 32
    assert not tok._fusable(X())


@pytest.mark.timeout(300)
def test_host_code_is_clean_under_address_sanitizer():
    """SURVEY §5 (sanitizers) / VERDICT r3 #9: the host-only native code -- curve tables, pixel tables, tile descriptors, the
    error path -- built with -fsanitize=address,undefined and driven with exactly-sized heap buffers
    (csrc/hostcheck/host_check.cpp; `make asan`).  GPU ASan is not available on this pool; kernels are covered by the
    parity tests instead."""
    import shutil
    import subprocess
    if shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no toolchain")
    csrc = os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd", "csrc")
    out = subprocess.run(["make", "-s", "-C", csrc, "asan"], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "host_check ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_tile_descriptor_errors_are_negative_not_counts():
    """sfcvit_tile_descriptors returns a COUNT on success, so its errors must be < 0 (found by the sanitizer driver: the
    error paths returned +SFCVIT_EINVAL = 1, which reads as "one int32 written")."""
    import ctypes
    import numpy as np
    from sfcvit import _lib
    pix = np.arange(4 * 256, dtype=np.int32)
    out = np.zeros(8, dtype=np.int32)
    n = _lib.lib.sfcvit_tile_descriptors(ctypes.c_void_p(pix.ctypes.data), 4, 256, 32, ctypes.c_void_p(out.ctypes.data), 8)
    assert n < 0 and b"capacity" in _lib.lib.sfcvit_last_error()
