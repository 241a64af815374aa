#!/usr/bin/env bash
# Launcher in the spirit of the reference's ./train.sh (website reprod/index.astro:238-264): GPU list and rendezvous port
# come from the environment, one process per GPU, RCCL over xGMI.  The reference guide sets CUDA_VISIBLE_DEVICES
# (reprod/index.astro:238); ROCm honours that name too, and HIP_VISIBLE_DEVICES wins when both are set.
set -euo pipefail
: "${HIP_VISIBLE_DEVICES:=${CUDA_VISIBLE_DEVICES:-0}}"
: "${MASTER_PORT:=29500}"
unset CUDA_VISIBLE_DEVICES              # (both set would be intersected by the runtime: one list only)
export HIP_VISIBLE_DEVICES HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(echo "$HIP_VISIBLE_DEVICES" | tr ',' '\n' | grep -c .)
cd "$(dirname "$0")"
if [ "${GDKVM_TRAIN_SH_DRY_RUN:-0}" = "1" ]; then      # (tests: print what would be launched)
  echo "HIP_VISIBLE_DEVICES=$HIP_VISIBLE_DEVICES NGPU=$NGPU MASTER_PORT=$MASTER_PORT"
  exit 0
fi
if [ "$NGPU" -gt 1 ]; then
  exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 --master-port "$MASTER_PORT" \
       train.py --config config/config_gdkvm_01.yaml "$@"
else
  exec python train.py --config config/config_gdkvm_01.yaml "$@"
fi
