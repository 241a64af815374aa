#!/usr/bin/env python3
"""Per-step kernel table of the STEADY STATE of a profiled bench run, from rocprofv3's kernel_trace.csv: MIOpen's find
mode runs trial and reference kernels (naive_conv, 36 ms each) during warm-up, which swamp the --stats summary; this takes
the last K forward steps (a step ends with upsample_argmax_dice_kernel) and averages over them.
usage: python profiles/steady_state.py <kernel_trace.csv> <out.csv> "<command line that was profiled>" [K=10] [end-marker] [sequence.csv | -] [markers-per-step=1]
end-marker: substring of the kernel that ends a step (default upsample_argmax_dice; training: multi_tensor = the fused AdamW
kernels; runs of marker kernels closer than 8 dispatches count as one step end).  markers-per-step: a forward that runs its batch as n
groups of clips on n streams (round 5) ends with n marker kernels, interleaved with the slower group's last kernels: every n-th marker ends
a step; the header then also gives the steps' wall span (first start to last end), which is what overlaps.  markers-per-step = q (round 6,
forwards IN FLIGHT: whole-batch graphs replayed in turn on several host streams, so the kernels of consecutive steps interleave in time):
the trace is split by Queue_Id first, each queue's steps are found as for 1, and the header gives the wall time per step of the window
(first start to last end of the steps taken, over their number) -- what the bench's clock sees."""
import collections
import csv
import sys

from summarize import short


def main():
    src, dst, cmd = sys.argv[1:4]
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    marker = sys.argv[5] if len(sys.argv) > 5 else "upsample_argmax_dice"
    per_queue = len(sys.argv) > 7 and sys.argv[7] == "q"
    per_step = 1 if per_queue else int(sys.argv[7]) if len(sys.argv) > 7 else 1
    rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))
    if per_queue:
        return in_flight(rows, dst, cmd, K, marker)
    ends = []
    if per_step > 1:
        # every stream of a step ends with its marker kernel, so the step is complete at its per_step-th marker
        seen = 0
        for i, r in enumerate(rows):
            if marker in r["Kernel_Name"]:
                seen += 1
                if seen % per_step == 0:
                    ends.append(i)
    else:
        for i, r in enumerate(rows):
            if marker in r["Kernel_Name"]:
                if ends and i - ends[-1] <= 8:
                    ends[-1] = i
                else:
                    ends.append(i)
    # the bench's roofline timing launches come after the timed steps: a forward step contains convolutions
    steps = [(ends[i - 1] + 1, ends[i] + 1) for i in range(1, len(ends))
             if any("conv" in r["Kernel_Name"] or "igemm" in r["Kernel_Name"] for r in rows[ends[i - 1] + 1:ends[i] + 1])]
    if per_step > 1:                                        # (the bench's kernel-timing legs launch marker kernels too and can shift the pairing:
        lens = collections.Counter(hi - lo for lo, hi in steps)   #  a forward step is one of the many segments of the most common length)
        common = lens.most_common(1)[0][0]
        steps = [(lo, hi) for lo, hi in steps if hi - lo == common]
    steps = steps[-K:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for lo, hi in steps:
        for r in rows[lo:hi]:
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values())
    span = sum(max(int(r["End_Timestamp"]) for r in rows[lo:hi]) - int(rows[lo]["Start_Timestamp"]) for lo, hi in steps)
    if len(sys.argv) > 6 and sys.argv[6] != "-":                                   # the dispatch sequence of the last step: name, grid, workgroup, duration, gap before it
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
                f"sum of kernel durations per step = {tot / len(steps) / 1e6:.3f} ms"
                + (f"; wall span of a step (kernels of {per_step} streams overlap) = {span / len(steps) / 1e6:.3f} ms" if per_step > 1 else "") + "\n")
        f.write("Name,CallsPerStep,MicrosecondsPerStep,Percentage\n")
        for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{n},{c / len(steps):.1f},{d / len(steps) / 1e3:.1f},{100 * d / tot:.2f}\n")


def in_flight(rows, dst, cmd, K, marker):
    queues = collections.defaultdict(list)
    for r in rows:
        queues[r["Queue_Id"]].append(r)
    steps = []                                              # (queue, rows of one step)
    for q, qr in queues.items():
        ends = [i for i, r in enumerate(qr) if marker in r["Kernel_Name"]]
        segs = [qr[ends[i - 1] + 1:ends[i] + 1] for i in range(1, len(ends))]
        segs = [sg for sg in segs if any("conv" in r["Kernel_Name"] or "igemm" in r["Kernel_Name"] for r in sg)]
        steps += [(q, sg) for sg in segs]
    common = collections.Counter(len(sg) for _, sg in steps).most_common(1)[0][0]
    steps = sorted(((q, sg) for q, sg in steps if len(sg) == common), key=lambda t: int(t[1][-1]["End_Timestamp"]))
    steps = steps[-(K + 1):-1] if len(steps) > K + 1 else steps        # (the very last forward has no successor overlapping it)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for _, sg in steps:
        for r in sg:
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values())
    n = len(steps)
    ends = [int(sg[-1]["End_Timestamp"]) for _, sg in steps]
    pace = (ends[-1] - ends[0]) / (n - 1) if n > 1 else 0.0               # step completions per unit of time = what the bench's clock sees
    own = sum(int(sg[-1]["End_Timestamp"]) - int(sg[0]["Start_Timestamp"]) for _, sg in steps) / n
    with open(dst, "w") as f:
        f.write(f"# {cmd}\n# steady state: mean over the last {n} forward steps of the kernel trace, found per hardware queue "
                f"({len(set(q for q, _ in steps))} queues: whole-batch graphs replayed in turn on as many streams, consecutive steps overlap); "
                f"sum of kernel durations per step = {tot / n / 1e6:.3f} ms; one step first kernel to last = {own / 1e6:.3f} ms; "
                f"a step completes every {pace / 1e6:.3f} ms\n")
        f.write("Name,CallsPerStep,MicrosecondsPerStep,Percentage\n")
        for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{name},{c / n:.1f},{d / n / 1e3:.1f},{100 * d / tot:.2f}\n")


if __name__ == "__main__":
    main()
