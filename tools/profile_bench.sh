#!/bin/bash
# The profiles bench.py's numbers are checked against (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command      -> $OUT/kernel_stats.csv
#   2. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (never combined with a trace domain other than the kernel
#      trace; MI355X_MICROARCH.md "HBM")                                   -> tools/traffic_from_pmc.py -> profiles/traffic.json
# usage: tools/profile_bench.sh <out dir under gpurun_out> <build tag> [workload]
set -e
export TMPDIR=/tmp
OUT=gpurun_out/${1:-prof}
TAG=${2:-build}
WL=${3:-vit_b16_224_hilbert}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/write.log 2>&1
python3 tools/traffic_from_pmc.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) "$TAG" $WL > $OUT/traffic.txt
cp profiles/traffic.json $OUT/traffic.json
tail -3 $OUT/stats.log | cut -c1-200
