#!/bin/bash
# final round-4 measurements (GPU box, repo root): default bench line, profiles (kernel stats + PMC traffic), other configs
set -o pipefail
python bench.py > gpurun_out/r4_bench_default.log 2>&1 && grep '^{' gpurun_out/r4_bench_default.log | tail -1 > gpurun_out/bench_b256_r4.json && cut -c1-400 gpurun_out/bench_b256_r4.json
bash tools/profile_bench.sh r4_final r4-final && echo "profiles done"
python bench.py --graph --no-cpu-baseline > gpurun_out/r4_bench_graph.log 2>&1; grep '^{' gpurun_out/r4_bench_graph.log | tail -1 > gpurun_out/bench_b256_r4_graph.json; cut -c1-200 gpurun_out/bench_b256_r4_graph.json
python bench.py --workload vit_l16_384_hilbert --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/r4_bench_l.log 2>&1; grep '^{' gpurun_out/r4_bench_l.log | tail -1 > gpurun_out/bench_vit_l16_384_hilbert_r4.json; cut -c1-200 gpurun_out/bench_vit_l16_384_hilbert_r4.json
python bench.py --workload vit_tiny16_32_hilbert --no-cpu-baseline > gpurun_out/r4_bench_t.log 2>&1; grep '^{' gpurun_out/r4_bench_t.log | tail -1 > gpurun_out/bench_vit_tiny16_32_hilbert_r4.json; cut -c1-200 gpurun_out/bench_vit_tiny16_32_hilbert_r4.json
