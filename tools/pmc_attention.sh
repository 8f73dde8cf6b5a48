#!/bin/bash
# SQ counters of the attention kernels (two passes of <= 8 SQ counters each); run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_attn}
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $OUT/p1 -- python3 tools/bench_attention.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC \
  --kernel-trace --output-format csv -d $OUT/p2 -- python3 tools/bench_attention.py > $OUT/p2.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    files = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "attn" not in k:
                continue
            k = k.replace("void ", "").replace("sfcvit::(anonymous namespace)::", "").split("(")[0]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(agg):
        print(p, k, {c: round(sum(v) / len(v)) for c, v in agg[k].items()})
PY
