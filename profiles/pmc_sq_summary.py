#!/usr/bin/env python3
"""Per-kernel MFMA-busy and wave-state fractions from one rocprofv3 SQ counter pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE).
usage: python profiles/pmc_sq_summary.py <counter_collection.csv> <out.csv> "<cmd>"
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is
summed over the 8 XCDs; MI355X_MICROARCH.md, DVFS note).  SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles: 16 per
v_mfma_f32_16x16x32_bf16, 32 per v_mfma_f32_16x16x4_f32.  The wave-state columns are shares of SQ_WAVE_CYCLES."""
import collections
import csv
import os
import re
import sys
import time

from summarize import short

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stamp() -> str:
    """Header line tying the summary to the kernel sources it was measured on: bench.py quotes a summary only while the
    hash still equals gdkvm_amd.build.source_hash(), and picks the newest summary by `collected`, not by file name."""
    from gdkvm_amd.build import source_hash
    return f"# scan_source_hash: {source_hash()} collected: {time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime())}\n"

# kernels kept in the summary: the product's own (hot path, epilogues, convolutions) -- MIOpen's find-mode trial kernels are dropped
KEEP = r"gdr_|kpff|proj_|argmax|conv3x3_|conv_igemm|grouped_conv|upsample|bias_|stem_|gate_logits|maxpool|bn_|seg_loss|head_|block_"


def main():
    src, dst, cmd = sys.argv[1:4]
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(src)):
        d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(dst, "w") as f:
        f.write(stamp())
        f.write(f"# {cmd}\n# per-launch means\n")
        f.write("Kernel,launches,kernel_cycles,SQ_VALU_MFMA_BUSY_CYCLES,mfma_busy_frac,wait_any_share,wait_inst_share,active_inst_share\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))):
            if not re.search(KEEP, k):
                continue
            m = {c: sum(x) / len(x) for c, x in v.items()}
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            wc = m["SQ_WAVE_CYCLES"] or 1.0
            f.write(f"{k},{len(v['GRBM_GUI_ACTIVE'])},{cyc:.0f},{m['SQ_VALU_MFMA_BUSY_CYCLES']:.0f},"
                    f"{m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.4f},{m['SQ_WAIT_ANY'] / wc:.3f},"
                    f"{m['SQ_WAIT_INST_ANY'] / wc:.3f},{m['SQ_ACTIVE_INST_ANY'] / wc:.3f}\n")


if __name__ == "__main__":
    main()
