#!/usr/bin/env python3
"""Per-step kernel table of the STEADY STATE of a profiled bench run, from rocprofv3's kernel_trace.csv: MIOpen's find
mode runs trial and reference kernels (naive_conv, 36 ms each) during warm-up, which swamp the --stats summary; this takes
the last K forward steps (a step ends with upsample_argmax_dice_kernel) and averages over them.
usage: python profiles/steady_state.py <kernel_trace.csv> <out.csv> "<command line that was profiled>" [K=10] [end-marker] [sequence.csv]
end-marker: substring of the kernel that ends a step (default upsample_argmax_dice; training: multi_tensor = the fused AdamW
kernels; runs of marker kernels closer than 8 dispatches count as one step end)."""
import collections
import csv
import sys

from summarize import short


def main():
    src, dst, cmd = sys.argv[1:4]
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    marker = sys.argv[5] if len(sys.argv) > 5 else "upsample_argmax_dice"
    rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))
    ends = []
    for i, r in enumerate(rows):
        if marker in r["Kernel_Name"]:
            if ends and i - ends[-1] <= 8:
                ends[-1] = i
            else:
                ends.append(i)
    # the bench's roofline timing launches come after the timed steps: a forward step contains convolutions
    steps = [(ends[i - 1] + 1, ends[i] + 1) for i in range(1, len(ends))
             if any("conv" in r["Kernel_Name"] or "igemm" in r["Kernel_Name"] for r in rows[ends[i - 1] + 1:ends[i] + 1])]
    steps = steps[-K:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for lo, hi in steps:
        for r in rows[lo:hi]:
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values())
    if len(sys.argv) > 6:                                   # the dispatch sequence of the last step: name, grid, workgroup, duration, gap before it
        lo, hi = steps[-1]
        with open(sys.argv[6], "w") as f:
            f.write(f"# {cmd}\n# dispatch sequence of the last steady-state step: index, microseconds, idle microseconds before it, grid, workgroup, kernel\n")
            for i in range(lo, hi):
                r = rows[i]
                gap = (int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i > lo else 0.0
                f.write(f"{i - lo},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f},{gap:.1f},{r.get('Grid_Size_X', '?')},{r.get('Workgroup_Size_X', '?')},"
                        f"{short(r['Kernel_Name'])}\n")
    with open(dst, "w") as f:
        f.write(f"# {cmd}\n# steady state: mean over the last {len(steps)} forward steps of the kernel trace; "
                f"sum of kernel durations per step = {tot / len(steps) / 1e6:.3f} ms\n")
        f.write("Name,CallsPerStep,MicrosecondsPerStep,Percentage\n")
        for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{n},{c / len(steps):.1f},{d / len(steps) / 1e3:.1f},{100 * d / tot:.2f}\n")


if __name__ == "__main__":
    main()
