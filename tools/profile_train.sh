#!/bin/bash
# Steady-state per-step kernel table of the training step (BASELINE.json configs[3] at the per-GPU shape: 16 clips x 32 frames of 112x112, bf16
# autocast, AdamW) into gpurun_out/<tag>_train_cfg4_steady_state.csv; copy into profiles/ to be judged.   usage: bash tools/profile_train.sh <tag>
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_train
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="python3 bench.py --mode train --steps 12 --warmup 5"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- $CMD > "$OUT/trace.log" 2>&1
cd profiles
python3 steady_state.py "$(ls ../$OUT/*kernel_trace.csv | head -1)" ../gpurun_out/${TAG}_train_cfg4_steady_state.csv "rocprofv3 --kernel-trace -- $CMD" 8 multi_tensor ../gpurun_out/${TAG}_train_cfg4_sequence.csv
cd ..
head -40 gpurun_out/${TAG}_train_cfg4_steady_state.csv
tail -3 "$OUT/trace.log"
