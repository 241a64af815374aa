#!/usr/bin/env python3
"""Diagnostic: one of the timing tools against two builds of the library, same box, alternating processes.
  python tools/ab_tool.py tools/argmax_probe.py tools/_abl/head.so gdkvm_amd/libgdkvm_hip.so [rounds]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    from gdkvm_amd import ops
    ops._SO = os.path.abspath(sys.argv[3])
    import runpy
    sys.argv = [sys.argv[2]]
    runpy.run_path(os.path.join(ROOT, sys.argv[0]), run_name="__main__")
else:
    script, libs, rounds = sys.argv[1], sys.argv[2:4], int(sys.argv[4]) if len(sys.argv) > 4 else 2
    for _ in range(rounds):
        for lib in libs:
            print("==", lib, flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", script, lib])
