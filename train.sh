#!/usr/bin/env bash
# Launcher in the spirit of the reference's ./train.sh (website reprod/index.astro:238-264): GPU list and rendezvous port
# come from the environment, one process per GPU, RCCL over xGMI.
set -euo pipefail
: "${HIP_VISIBLE_DEVICES:=0}"
: "${MASTER_PORT:=29500}"
export HIP_VISIBLE_DEVICES HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(echo "$HIP_VISIBLE_DEVICES" | tr ',' '\n' | grep -c .)
cd "$(dirname "$0")"
if [ "$NGPU" -gt 1 ]; then
  exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 --master-port "$MASTER_PORT" \
       train.py --config config/config_gdkvm_01.yaml "$@"
else
  exec python train.py --config config/config_gdkvm_01.yaml "$@"
fi
