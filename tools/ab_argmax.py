#!/usr/bin/env python3
"""Diagnostic: tools/argmax_probe.py against another build of the library, same box, alternating processes.
  python tools/ab_argmax.py tools/_abl/head.so gdkvm_amd/libgdkvm_hip.so [rounds]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    from gdkvm_amd import ops
    ops._SO = os.path.abspath(sys.argv[2])
    import runpy
    runpy.run_path(os.path.join(ROOT, "tools", "argmax_probe.py"), run_name="__main__")
else:
    libs, rounds = sys.argv[1:3], int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for _ in range(rounds):
        for lib in libs:
            print("==", lib, flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", lib])
