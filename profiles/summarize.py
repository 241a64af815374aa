#!/usr/bin/env python3
"""Shorten a rocprofv3 *_kernel_stats.csv into a readable, committed summary.
usage: python profiles/summarize.py <kernel_stats.csv> <out.csv> "<command line that was profiled>" [drop-regex]
Rows whose kernel name matches drop-regex (e.g. MIOpen's find-mode trial kernels of the warm-up) are left out and the
percentages recomputed over the rest; the header line says so."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("at::native::", "")
    m = re.match(r"([A-Za-z_0-9:]+)(<[^(]{0,40})?", name)
    s = (m.group(1) + (m.group(2) or "")) if m else name
    if "elementwise_kernel" in s or "distribution" in s:
        inner = re.search(r"(bfloat16_copy|bfloat16tofloat32_copy|direct_copy|launch_clamp|CUDAFunctor_add|CUDAFunctorOnSelf_add|normal_kernel)", name)
        if inner:
            s = s.split("<")[0] + "[" + inner.group(1) + "]"
    return s[:100].replace(",", ";")


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    drop = re.compile(sys.argv[4]) if len(sys.argv) > 4 else None
    rows = list(csv.DictReader(open(src)))
    if drop:
        kept = [r for r in rows if not drop.search(r["Name"])]
        gone = sum(float(r["TotalDurationNs"]) for r in rows) - sum(float(r["TotalDurationNs"]) for r in kept)
        tot = sum(float(r["TotalDurationNs"]) for r in kept)
        for r in kept:
            r["Percentage"] = f"{100 * float(r['TotalDurationNs']) / tot:.2f}"
        rows = kept
        cmd += f"   [rows matching /{sys.argv[4]}/ dropped: {gone / 1e6:.1f} ms]"
    with open(dst, "w") as f:
        if "hotpath" in dst or "scan_" in dst:               # tie the scan kernels' durations to the sources they were measured on (bench.py quotes them)
            import os
            import time
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from gdkvm_amd.build import source_hash
            f.write(f"# scan_source_hash: {source_hash()} collected: {time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime())}\n")
        f.write(f"# {cmd}\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},"
                    f"{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")


if __name__ == "__main__":
    main()
