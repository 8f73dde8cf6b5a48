#!/bin/bash
set -o pipefail
for nt in 0 1 0 1; do
  SFCVIT_ATTN_NT=$nt python bench.py --no-cpu-baseline --time-all-kernels > gpurun_out/r4_bench_attn_nt$nt.log 2>&1
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r4_bench_attn_nt$nt.log") if x.startswith("{")][-1])
r=d["roofline_detail"]
print("ATTN_NT=$nt", d["value"], d["ms_per_step"], "bwd", r["attention_bwd"]["avg_launch_ms"], "fwd", r["attention_fwd"]["avg_launch_ms"], "gemm frac", d["roofline"]["frac"])
PY
done
