#!/bin/bash
# Per-kernel durations of one gdkvm_scan_fwd at a given shape (rocprofv3 --kernel-trace --stats of tools/scan_only.py), summary into
# gpurun_out/<tag>_scan_<name>_kernel_stats.csv.     usage: bash tools/profile_scan_shape.sh <tag> <name> B T N Dv [iters]
set -e
TAG=$1; NAME=$2; B=$3; T=$4; N=$5; DV=$6; IT=${7:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_${NAME}
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="python3 tools/scan_only.py $B $T $N $DV $IT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- $CMD > "$OUT/trace.log" 2>&1
cd profiles
python3 summarize.py "$(ls ../$OUT/*kernel_stats.csv | head -1)" ../gpurun_out/${TAG}_scan_${NAME}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- $CMD   ($NAME: B=$B, T=$T, N=$N, Dv=$DV, bf16)"
cd ..
cat gpurun_out/${TAG}_scan_${NAME}_kernel_stats.csv
