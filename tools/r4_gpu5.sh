#!/bin/bash
# in-step A/B of the LayerNorm forward kernels (rocprofv3 kernel stats of the default bench, same box)
export TMPDIR=/tmp
for v in "1 0" "1 1" "0 0" "1 0" "1 1"; do
  set -- $v
  tag=t$1_n$2
  rm -rf gpurun_out/ln_ab_$tag
  SFCVIT_LN_FWD_TWO_ROWS=$1 SFCVIT_LN_NT=$2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ln_ab_$tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/ln_ab_$tag.log 2>&1
  f=$(find gpurun_out/ln_ab_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$tag" "$(grep '^{' gpurun_out/ln_ab_$tag.log | tail -1 | cut -c100-175)" <<'PY'
import csv,sys
out=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'ln_fwd' in r['Name'] or 'gemm8p_kernel<8, 0' in r['Name'] or 'gemm8p_kernel<8, 35' in r['Name']:
        out.append(r['Name'].split('::')[-1][:26]+" avg us %.1f" % (float(r['AverageNs'])/1e3))
print(sys.argv[2], sys.argv[3], " | ".join(out))
PY
done
