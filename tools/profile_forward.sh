#!/bin/bash
# Round-6 counter evidence for the kernels that dominate the headline forward (the convolutions either side of the memory path): kernel-trace
# stats of tools/forward_only.py plus the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ wave-state counters) in their own runs (MI355X_MICROARCH.md:
# the TCC counters cannot share a pass; the program stands directly after `--`).  Summaries land in gpurun_out/<tag>_forward_cfg2_*.csv;
# copy the ones to be judged into profiles/.        usage: bash tools/profile_forward.sh <tag> [iters]
set -e
TAG=${1:-r06}
IT=${2:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_forward
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="python3 tools/forward_only.py $IT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o t -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o t -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/sq" -o t -- $CMD > "$OUT/sq.log" 2>&1
cd profiles
python3 summarize.py "$(ls ../$OUT/trace/*kernel_stats.csv | head -1)" ../gpurun_out/${TAG}_forward_cfg2_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- $CMD   (cfg2: 16 clips x 32 frames x 112x112, bf16, eager launches on one stream)"
python3 pmc_summary.py "$(ls ../$OUT/fetch/*counter_collection.csv | head -1)" "$(ls ../$OUT/write/*counter_collection.csv | head -1)" \
    ../gpurun_out/${TAG}_forward_cfg2_pmc_hbm.csv "rocprofv3 --pmc FETCH_SIZE -- $CMD  ;  rocprofv3 --pmc WRITE_SIZE -- $CMD   (separate passes; cfg2, bf16)"
python3 pmc_sq_summary.py "$(ls ../$OUT/sq/*counter_collection.csv | head -1)" ../gpurun_out/${TAG}_forward_cfg2_pmc_sq.csv \
    "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- $CMD   (cfg2, bf16)"
cd ..
cat gpurun_out/${TAG}_forward_cfg2_kernel_stats.csv gpurun_out/${TAG}_forward_cfg2_pmc_hbm.csv gpurun_out/${TAG}_forward_cfg2_pmc_sq.csv
