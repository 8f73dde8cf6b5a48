#!/bin/bash
# Launcher with the role of the reference's run_vit.sh (a SLURM wrapper around `python main.py`): one process per
# GPU of this node over RCCL/xGMI.  Usage: ./run_vit.sh [N_GPUS] [main.py arguments...]
set -euo pipefail
cd "$(dirname "$0")"
NGPU="${1:-$(python3 -c 'import torch; print(max(1, torch.cuda.device_count()))')}"
shift || true
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC (RCCL / shared device memory across processes)
if [ "$NGPU" -le 1 ]; then
    exec python3 main.py "$@"
fi
exec python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 \
    --master-port "${MASTER_PORT:-29500}" main.py "$@"
