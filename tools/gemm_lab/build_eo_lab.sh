#!/bin/bash
# Builds the round-4 epilogue experiments as variant libraries (tools/probe/*.so, loaded through SFCVIT_LIB):
#   lib_eo.so    gemm8p_eo_lab.hip: schedules 0 / 1 as the product + schedule 2 (overlapped epilogue), SFCVIT_GEMM_SCHED picks
#   lib_regx.so  the same source with -DSFCVIT_GEMM_REG_EXCHANGE: schedule 1 with v_permlane16/32_swap instead of the LDS patch
# The product objects must exist (make -C csrc).  The kernel name the library reports carries the schedule's LOW BIT as
# true / false (the product's formatter): schedule 2 prints as <NI, MASK, false>.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/space-filling-curves-for-vision-transformers_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=fast -mllvm -amdgpu-atomic-optimizer-strategy=None -I$CS -x hip"
OTHERS=$(ls $CS/build/*.o | grep -v "build/gemm8p.o")
mkdir -p $ROOT/tools/probe /tmp/eo_lab
/opt/rocm/bin/hipcc $FLAGS -c $ROOT/tools/gemm_lab/gemm8p_eo_lab.hip -o /tmp/eo_lab/eo.o &
/opt/rocm/bin/hipcc $FLAGS -DSFCVIT_GEMM_REG_EXCHANGE -c $ROOT/tools/gemm_lab/gemm8p_eo_lab.hip -o /tmp/eo_lab/regx.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/lib_eo.so $OTHERS /tmp/eo_lab/eo.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/lib_regx.so $OTHERS /tmp/eo_lab/regx.o
ls -la $ROOT/tools/probe/lib_eo.so $ROOT/tools/probe/lib_regx.so
