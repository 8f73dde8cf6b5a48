#!/bin/bash
# Matrix-pipe occupancy of every kernel of the benched step from hardware counters (one --pmc pass with the kernel trace,
# no other trace domain): SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over the chip: 16 per
# v_mfma_f32_16x16x32_bf16) against the nominal 2.4 GHz x 1 024 SIMDs, and SQ_INSTS_VALU_MFMA_MOPS_BF16 (512-FLOP units) as
# a hardware count of the FLOPs the kernels are credited with.
#   usage (GPU box, repo root): tools/pmc_mfma.sh [out dir under gpurun_out]
set -e
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc_mfma}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $OUT/p \
  -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/p.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
cc = glob.glob(f"{out}/p/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(f"{out}/p/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(cc)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("sfcvit::(anonymous namespace)::", "").replace("(anonymous namespace)::", ""))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[k]:
        seen[k].add(r["Dispatch_Id"])
        agg[k]["ns"] += dur.get(r["Dispatch_Id"], 0)
print(f"{'kernel':52s} {'launches':>8s} {'avg us':>8s} {'MFMA busy cycles / (1024 SIMDs x ns x 2.4 GHz)':>48s} {'TFLOP/s from MOPS':>18s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    n = len(seen[k])
    busy, ns = v["SQ_VALU_MFMA_BUSY_CYCLES"], v["ns"]
    f2 = busy / (1024 * ns * 2.4)
    tf = v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) * 512 / ns / 1e3
    print(f"{k[:52]:52s} {n:8d} {ns / n / 1e3:8.1f} {f2:48.3f} {tf:18.1f}")
PY
