#!/bin/bash
# Steady-state per-step kernel table of the headline forward (rocprofv3 --kernel-trace of bench.py without its extra legs), into
# gpurun_out/<tag>_bench_cfg2_steady_state.csv; copy into profiles/ to be judged.     usage: bash tools/profile_bench.sh <tag>
# last argument of steady_state.py: q = forwards in flight (the shipped timed loop: steps found per hardware queue); 2 with GDKVM_BENCH_IN_FLIGHT=1
# (one graph at a time, two groups of clips on two streams inside): GDKVM_STEADY_MODE=2 GDKVM_BENCH_IN_FLIGHT=1 bash tools/profile_bench.sh <tag>
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_bench
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="python3 bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-user-path-legs --train-steps 0 --kernel-iters 3"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- $CMD > "$OUT/trace.log" 2>&1
cd profiles
python3 steady_state.py "$(ls ../$OUT/*kernel_trace.csv | head -1)" ../gpurun_out/${TAG}_bench_cfg2_steady_state.csv "rocprofv3 --kernel-trace -- $CMD" 10 upsample_argmax_dice - ${GDKVM_STEADY_MODE:-q}
cd ..
cat gpurun_out/${TAG}_bench_cfg2_steady_state.csv
