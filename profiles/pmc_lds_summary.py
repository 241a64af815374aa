#!/usr/bin/env python3
"""Per-kernel LDS activity from one rocprofv3 SQ counter pass
(--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE).
usage: python profiles/pmc_lds_summary.py <counter_collection.csv> <out.csv> "<cmd>"
MI355X_MICROARCH.md §LDS: SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra cycles conflicts cost (both summed
over the chip's 256 CUs).  Derived columns:
  lds_busy_frac   SQ_LDS_IDX_ACTIVE / (kernel cycles x 256 CUs), kernel cycles = GRBM_GUI_ACTIVE / 8 -- the share of the kernel during
                  which a CU's LDS array was working (1.0 = LDS-bound)
  conflict_share  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  lds_inst_share, lds_stall_share   SQ_ACTIVE_INST_LDS, SQ_WAIT_INST_LDS as shares of SQ_WAVE_CYCLES (quad-cycle units both)"""
import collections
import csv
import re
import sys

from pmc_sq_summary import KEEP, stamp
from summarize import short


def main():
    src, dst, cmd = sys.argv[1:4]
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(src)):
        d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(dst, "w") as f:
        f.write(stamp())
        f.write(f"# {cmd}\n# per-launch means\n")
        f.write("Kernel,launches,kernel_cycles,SQ_INSTS_LDS,SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT,lds_busy_frac,conflict_share,lds_inst_share,lds_stall_share\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1].get("SQ_LDS_IDX_ACTIVE", [0]))):
            if not re.search(KEEP, k):
                continue
            m = collections.defaultdict(float, {c: sum(x) / len(x) for c, x in v.items()})
            cyc = m["GRBM_GUI_ACTIVE"] / 8 or 1.0
            wc = m["SQ_WAVE_CYCLES"] or 1.0
            idx = m["SQ_LDS_IDX_ACTIVE"]
            f.write(f"{k},{len(v['GRBM_GUI_ACTIVE'])},{cyc:.0f},{m['SQ_INSTS_LDS']:.0f},{idx:.0f},{m['SQ_LDS_BANK_CONFLICT']:.0f},"
                    f"{idx / (cyc * 256):.4f},{m['SQ_LDS_BANK_CONFLICT'] / (idx or 1.0):.4f},{m['SQ_ACTIVE_INST_LDS'] / wc:.3f},"
                    f"{m['SQ_WAIT_INST_LDS'] / wc:.3f}\n")


if __name__ == "__main__":
    main()
