#!/usr/bin/env python3
"""Per-kernel resource usage of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage): registers, spills, scratch, LDS.
usage: kres.py gdr_scan.hip [name filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gdkvm_amd import build  # noqa: E402

src = os.path.join(build.CSRC, sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = [build.HIPCC] + build.FLAGS + build.EXTRA_FLAGS.get(sys.argv[1], []) + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'VGPR':>5s} {'AGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>7s}  kernel")
for r in rows:
    if flt in r["name"]:
        print(f"{r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>6s} {r.get('SGPRs Spill', '?'):>6s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} {r.get('LDS Size [bytes/block]', '?'):>7s}  {r['name'][:150]}")
