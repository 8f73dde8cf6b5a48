#!/bin/bash
# round 4: per-element dropout hash -- tests, then attention and GEMM A/B against the round-3 library (tools/probe/lib_head.so)
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
echo "== dropout / attention / gemm tests"; timeout -k 10 900 python -m pytest tests/test_dropout_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -5
echo "== attention, new hash"; timeout -k 10 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_attn_newhash.txt
echo "== attention, round-3 library, same box"; SFCVIT_LIB=$PWD/tools/probe/lib_head.so timeout -k 10 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_attn_head.txt
echo "== gemm, new hash"; AB_SCHEDS=1 timeout -k 10 300 python tools/gemm_lab/ab_sched.py 2>&1 | grep -v amdgpu.ids
echo "== gemm, round-3 library"; SFCVIT_LIB=$PWD/tools/probe/lib_head.so AB_SCHEDS=1 timeout -k 10 300 python tools/gemm_lab/ab_sched.py 2>&1 | grep -v amdgpu.ids
