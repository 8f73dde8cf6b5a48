#!/bin/bash
# round 4: the two epilogue experiments, correctness then A/B on one box (profiles/r4/gemm_epilogue_overlap_ab.txt)
#   product            exchange through the LDS patch, epilogue between tiles (schedule 1)
#   SFCVIT_GEMM_SCHED=2  overlapped epilogue (row chores in the load sections of the boundary k-tiles; register exchange)
#   lib_regx.so        schedule 1 with the register exchange (v_permlane16/32_swap) instead of the LDS patch
set -o pipefail
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== gemm tests, schedule 1"; SFCVIT_GEMM_SCHED=1 timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3
echo "== gemm tests, schedule 2 (overlapped epilogue)"; SFCVIT_GEMM_SCHED=2 timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -8
echo "== gemm tests, register exchange build"; SFCVIT_LIB=$PWD/tools/probe/lib_regx.so timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3
for i in 1 2; do python tools/gemm_lab/dbg_sched.py 9472 1792 256 8 2>&1 | grep -v amdgpu.ids | grep trial; done
echo "== A/B schedule 1 vs 2 (product library)"; timeout -k 10 300 python tools/gemm_lab/ab_sched.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_ab_sched.txt
echo "== schedule 1 with the register exchange, same box"; SFCVIT_LIB=$PWD/tools/probe/lib_regx.so AB_SCHEDS=1 timeout -k 10 300 python tools/gemm_lab/ab_sched.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_ab_regx.txt
