"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol include/gdkvm.h
declares, and the host wrappers refuse to run without a device (no fallback path)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gdkvm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gdkvm_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    from gdkvm_amd import build, ops
    build.build()
    lib = ops.load()
    declared = _declared_symbols()
    assert declared, "no symbols parsed from include/gdkvm.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gdkvm.h but not exported"
        assert name in ops.SIGNATURES, f"{name} has no ctypes signature in gdkvm_amd/ops.py"
    assert lib.gdkvm_abi_version() == 1
    assert lib.gdkvm_scan_workspace_bytes(16, 32, 1, 49, 64, 256) == (16 * 32 * (64 * (6 * 64 + 256 + 1 + 16) + 64 * 96 + 64 * (64 + 32 + 256) + 256 // 4) + 16 * 16) * 4 + 1024 + 1024      # (+ gmax per frame and slice, + 2^e per clip and slice)


def test_no_cpu_fallback():
    from gdkvm_amd import ops
    x = torch.zeros(1, 1, 4, 1, 64)
    with pytest.raises(ops.GdkvmError, match="device"):
        ops.scan_fwd(x, x, torch.zeros(1, 1, 4, 1, 16), torch.zeros(1, 1, 1), torch.zeros(1, 1, 4, 1))
    with pytest.raises(ops.GdkvmError, match="device"):
        ops.argmax_dice(torch.zeros(1, 2, 4, 4))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under gdkvm_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gdkvm_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{fn} imports oracle"
                assert "gdkvm_oracle" not in txt, f"{fn} references the oracle"


def test_host_side_of_the_library_is_clean_under_sanitizers():
    """The HOST halves of the HIP library -- workspace carving (csrc/gdr_ws.hpp), the segmented scan's and the normalizer's own carves, argument
    checks and error reporting of gdr_segmented / gdr_normalizer / gdr_step / gdkvm_api -- built with -fsanitize=address,undefined on the CPU
    (tests/host_san/: device code compiled as usual and never run; the kernel-heavy entry points they call are stubbed) and driven with
    workspaces malloc'ed at EXACTLY the size the library asks for, every carved region written end to end: a carve that outgrows its own size
    request is a heap-buffer-overflow report, not a silent overwrite on the GPU box.  Also pins the error codes of ~25 bad-argument calls."""
    import subprocess
    here = os.path.join(ROOT, "tests", "host_san")
    subprocess.check_call(["bash", os.path.join(here, "build.sh")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(here, "_build", "host_san")], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0 and "host_san: ok" in out.stdout, out.stdout + out.stderr
