#!/bin/bash
# gather variants: alone (warm / clean-cold / dirty-cold) and inside the step
set -o pipefail
for u in 1 2; do for nt in 1 3; do
  echo "== SFCVIT_GATHER_U=$u SFCVIT_GATHER_NT=$nt"
  SFCVIT_GATHER_U=$u SFCVIT_GATHER_NT=$nt python tools/bench_patch_embed.py > gpurun_out/r4_pe_u${u}_nt$nt.log 2>&1; grep "gather, tile" gpurun_out/r4_pe_u${u}_nt$nt.log
  SFCVIT_GATHER_U=$u SFCVIT_GATHER_NT=$nt python bench.py --no-cpu-baseline --time-all-kernels > gpurun_out/r4_bench_u${u}_nt$nt.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/r4_bench_u${u}_nt$nt.log") if x.startswith("{")][-1]
d=json.loads(l)
print("in step:", d["value"], d["ms_per_step"], {k:v for k,v in d.get("roofline_detail",{}).items() if "gather" in k})
PY
done; done
