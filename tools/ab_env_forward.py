#!/usr/bin/env python3
"""A/B of one environment switch on ONE GPU box: times the cfg2 inference forward in fresh processes, alternating the two settings
(forward times differ by a few per cent between boxes).  usage: ab_env_forward.py VAR valueA valueB [rounds=3]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from tools.ab_forward import one
    if sys.argv[1] == "one":
        one(os.path.join(ROOT, "gdkvm_amd", "libgdkvm_hip.so"))
    else:
        var, va, vb = sys.argv[1:4]
        for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 3):
            for v in (va, vb):
                print(f"{var}={v}: ", end="", flush=True)
                subprocess.check_call([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **{var: v}))
