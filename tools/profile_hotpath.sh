#!/bin/bash
# Collects the round's rocprofv3 evidence for the hot-path kernels on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of tools/hotpath_only.py, and the four PMC passes (FETCH_SIZE, WRITE_SIZE, SQ wave-state counters, SQ LDS counters) in their own
#   runs (MI355X_MICROARCH.md: the TCC counters cannot share a pass).  Summaries land in gpurun_out/<tag>_*.csv; copy the ones to
#   be judged into profiles/.        usage: bash tools/profile_hotpath.sh <tag> [iters]
set -e
TAG=${1:-r02}
IT=${2:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="python3 tools/hotpath_only.py $IT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o t -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o t -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/sq" -o t -- $CMD > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/lds" -o t -- $CMD > "$OUT/lds.log" 2>&1
cd profiles
python3 summarize.py "$(ls ../$OUT/trace/*kernel_stats.csv | head -1)" ../gpurun_out/${TAG}_hotpath_cfg2_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- $CMD   (cfg2 shapes, bf16)"
python3 pmc_summary.py "$(ls ../$OUT/fetch/*counter_collection.csv | head -1)" "$(ls ../$OUT/write/*counter_collection.csv | head -1)" \
    ../gpurun_out/${TAG}_hotpath_cfg2_pmc_hbm.csv "rocprofv3 --pmc FETCH_SIZE -- $CMD  ;  rocprofv3 --pmc WRITE_SIZE -- $CMD   (separate passes; cfg2 shapes, bf16)"
python3 pmc_sq_summary.py "$(ls ../$OUT/sq/*counter_collection.csv | head -1)" ../gpurun_out/${TAG}_hotpath_cfg2_pmc_sq.csv \
    "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- $CMD   (cfg2 shapes, bf16)"
python3 pmc_lds_summary.py "$(ls ../$OUT/lds/*counter_collection.csv | head -1)" ../gpurun_out/${TAG}_hotpath_cfg2_pmc_lds.csv \
    "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $CMD   (cfg2 shapes, bf16)"
cd ..
cat gpurun_out/${TAG}_hotpath_cfg2_pmc_lds.csv
cat gpurun_out/${TAG}_hotpath_cfg2_kernel_stats.csv gpurun_out/${TAG}_hotpath_cfg2_pmc_hbm.csv gpurun_out/${TAG}_hotpath_cfg2_pmc_sq.csv
