#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).
usage: python profiles/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.csv> "<cmd>"
Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; FETCH_SIZE reads
exactly 1/2 of the bytes of a wide (16 B/lane) coalesced read stream on gfx950, WRITE_SIZE is exact for 16 B/lane
stores; other access widths are uncalibrated.  The summary therefore reports the raw value and the x2-corrected
read bytes side by side (the truth for mixed-width kernels lies between them)."""
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


def load(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return d


def main():
    f, w, out, cmd = sys.argv[1:5]
    fe, wr = load(f, "FETCH_SIZE"), load(w, "WRITE_SIZE")
    with open(out, "w") as o:
        o.write(stamp())
        o.write(f"# {cmd}\n# per-launch means; KiB as reported by rocprofv3; read_x2 = FETCH_SIZE*2 (gfx950 wide-read correction)\n")
        o.write("Kernel,launches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes_raw,hbm_bytes_read_x2\n")
        for k in sorted(fe, key=lambda k: -sum(fe[k])):
            if not re.search(KEEP, k):
                continue
            fv = sum(fe[k]) / len(fe[k])
            wv = sum(wr[k]) / len(wr[k]) if wr.get(k) else 0.0
            o.write(f"{k},{len(fe[k])},{fv:.1f},{wv:.1f},{(fv + wv) * 1024:.0f},{(2 * fv + wv) * 1024:.0f}\n")


if __name__ == "__main__":
    main()
