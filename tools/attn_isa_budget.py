#!/usr/bin/env python3
"""Instruction budget of one step of attn_seq_bwd_fused_kernel<13, true> (ViT-B: N = 196, dropout on) from its ISA.

    python tools/attn_isa_budget.py > profiles/r4/attention_isa_budget.txt        (no GPU needed: hipcc cross-compiles)

Compiles csrc/attention_bwd_fused.hip for gfx950 with -DSFCVIT_ISA_MARKERS (comment marks around key_step / dq_step; the
marks are `asm volatile` with a memory clobber, so the marked build may schedule slightly differently from the product
build) and buckets the instructions between the marks.  A key wave's step covers 16 keys x 32 queries = 512 scores, 8 per
lane: VALU instructions per lane / 8 = VALU lane-ops per score (round 3's PMC figure: 34)."""
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "space-filling-curves-for-vision-transformers_amd", "csrc", "attention_bwd_fused.hip")
KERNEL = "attn_seq_bwd_fused_kernelILi13ELb1E"

HASH = {"v_xor_b32", "v_mul_lo_u32", "v_cndmask_b32", "v_cmp_ge_u32", "v_cmp_le_u32", "v_cmp_lt_u32", "v_cmp_gt_u32", "v_bitop3_b32"}


def bucket(ins):
    base = ins.rsplit("_e32", 1)[0].rsplit("_e64", 1)[0]
    if ins.startswith("v_mfma"):
        return "MFMA"
    if ins.startswith("v_exp"):
        return "exp2"
    if ins.startswith("ds_"):
        return "LDS"
    if ins.startswith("v_cvt_pk_bf16"):
        return "fp32 -> bf16 packing (v_cvt_pk_bf16_f32)"
    if base in HASH:
        return "dropout hash + keep / boundary selects (xor, mul_lo, cmp, cndmask)"
    if ins.startswith(("v_fma", "v_mul_f32", "v_sub_f32", "v_add_f32", "v_pk_mul", "v_pk_fma", "v_pk_add", "v_fmac", "v_mad_f")):
        return "fp arithmetic (scale, P, dS = P (dP - delta), sum_q dS)"
    if ins.startswith(("v_mov", "v_perm", "v_lshl_or", "v_and_or", "v_accvgpr", "v_bfi", "v_alignbit", "v_permlane", "v_swap")):
        return "moves / register packing"
    if ins.startswith("v_"):
        return "integer / LDS address arithmetic"
    if ins.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
        return "waits / nops"
    if ins.startswith("s_"):
        return "SALU / branches"
    if ins.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vector memory"
    return "other"


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "bwd.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-DSFCVIT_ISA_MARKERS", "-x", "hip",
                        "--cuda-device-only", "-S", SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and KERNEL in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    print(f"# {KERNEL.replace('ILi13ELb1E', '<13, true>')}: instructions between the ISA marks (hipcc -O3, gfx950, ROCm 7.2)")
    for region, per in (("key_step", "16 keys x 32 queries = 8 scores per lane"), ("dq_step", "dQ of 32 queries x 32 head columns from 224 keys")):
        b = next(i for i in range(start, end) if f"ISA_MARK {region} begin" in lines[i])
        e = next(i for i in range(b, end) if f"ISA_MARK {region} end" in lines[i])
        counts = collections.OrderedDict()
        for l in lines[b + 1:e]:
            t = l.strip()
            if not t or t[0] in ".;" or t.endswith(":"):
                continue
            ins = t.split()[0]
            counts.setdefault(bucket(ins), collections.Counter())[ins] += 1
        valu = sum(sum(c.values()) for k, c in counts.items() if k not in ("MFMA", "LDS", "waits / nops", "SALU / branches", "vector memory", "other"))
        print(f"\n## {region} (one wave, one step: {per})")
        for k, c in sorted(counts.items(), key=lambda kv: -sum(kv[1].values())):
            print(f"{sum(c.values()):5d}  {k:72s} " + ", ".join(f"{i} x{n}" for i, n in c.most_common(6)))
        print(f"VALU instructions (everything but MFMA / LDS / SALU / waits / vector memory): {valu}"
              + (f" = {valu / 8:.1f} per score" if region == "key_step" else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
