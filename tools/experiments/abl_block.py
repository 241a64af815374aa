#!/usr/bin/env python3
"""Experiment driver for tools/experiments/conv3x3_block64.hip (a stride-4 residual block as ONE kernel; withdrawn from the product, see
profiles/r06_e_fused_block_stamps.txt): builds it into tools/_abl/cb.so (+ cb_stamps.so with -DCB_DIAG), checks it torch.equal against the
two launches of the product's 64 -> 64 kernel, times both, prints the per-phase stamps.
  python tools/experiments/abl_block.py build        (here)          python tools/experiments/abl_block.py run [frames]   (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "gdkvm_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_abl")
SRC = os.path.join(ROOT, "tools", "experiments", "conv3x3_block64.hip")


def build():
    os.makedirs(OUT, exist_ok=True)
    api = os.path.join(CSRC, "_obj", "gdkvm_api.o")
    for name, flags in (("cb", []), ("cb_stamps", ["-DCB_DIAG"])):
        obj, so = os.path.join(OUT, name + ".o"), os.path.join(OUT, name + ".so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + flags + sys.argv[2:] +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, SRC, "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj, api])
        os.remove(obj)
        print("built", so, flush=True)


def _lib(name):
    lib = ctypes.CDLL(os.path.join(OUT, name + ".so"))
    lib.gdkvm_conv_block_bias_act.restype = ctypes.c_int
    lib.gdkvm_conv_block_bias_act.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    lib.gdkvm_last_error.restype = ctypes.c_char_p
    return lib


def run():
    import torch
    from gdkvm_amd import ops
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    cl = torch.channels_last
    torch.manual_seed(0)
    w1, w2 = ((torch.randn(64, 64, 3, 3, device="cuda") / 24).bfloat16().contiguous(memory_format=cl) for _ in range(2))
    b1, b2 = torch.randn(64, device="cuda") * 0.3, torch.randn(64, device="cuda") * 0.3
    p1, p2 = ops.conv3x3_pack_weights(w1), ops.conv3x3_pack_weights(w2)

    def block(lib, x):
        y = torch.empty_like(x)
        rc = lib.gdkvm_conv_block_bias_act(x.data_ptr(), p1.data_ptr(), b1.data_ptr(), p2.data_ptr(), b2.data_ptr(), y.data_ptr(),
                                           x.shape[0], 64, x.shape[2], x.shape[3], 1, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.gdkvm_last_error().decode()
        return y

    def two(x):
        return ops.conv_bias_act(ops.conv_bias_act(x, w1, b1, None, 1, 1, True, 4, p1), w2, b2, x, 1, 1, True, 4, p2)

    lib = _lib("cb")
    for case in [(512, 28, 28), (5, 28, 28), (3, 10, 28), (2, 7, 20), (9, 15, 13), (1, 1, 1), (8, 30, 28), (17, 28, 28), (300, 28, 28)]:
        x = torch.randn(case[0], 64, case[1], case[2], device="cuda").bfloat16().contiguous(memory_format=cl)
        y, want = block(lib, x), two(x)
        torch.cuda.synchronize()
        ok = torch.equal(y, want)
        print(f"  {case}: {'bit-identical' if ok else 'MISMATCH max|d| = %g' % (y.float() - want.float()).abs().max().item()}", flush=True)
        assert ok and torch.equal(block(lib, x), y)
    xs = [torch.randn(n, 64, 28, 28, device="cuda").bfloat16().contiguous(memory_format=cl) for _ in range(6)]

    def ev(fn, it=30, warm=5):
        for i in range(warm):
            fn(xs[i % 6])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(it):
            fn(xs[i % 6])
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / it * 1e3
    print(f"frames {n}: fused block {ev(lambda x: block(lib, x)):.1f} us   two launches {ev(two):.1f} us", flush=True)

    sl = _lib("cb_stamps")
    buf = torch.zeros(10 * 8 * 8, dtype=torch.int64, device="cuda")
    sl.gdkvm_cb_diag_buffer.argtypes = [ctypes.c_void_p]
    sl.gdkvm_cb_diag_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        block(sl, xs[0])
    torch.cuda.synchronize()
    t = buf.cpu().reshape(10, 8, 8)
    print("stamps (s_memtime, workgroup 0), per phase and wave; producers (waves 0-3): MFMA pass A | epilogue A | MFMA pass B | meet + fetch issue | epilogue B | Y (band landed)")
    print("                                                      consumers (waves 4-7): MFMA pass A | epilogue A | MFMA pass B | -                 | epilogue B | Y      | (phase)")
    for c in range(1, 9):
        for wv in (0, 2, 4, 6):
            r = t[c, wv]
            print(f"  phase {c} wave {wv}: " + " | ".join(f"{int(r[i + 1] - r[i]):6d}" for i in range(6)) + f" | ({int(r[6] - r[0]):6d})")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
