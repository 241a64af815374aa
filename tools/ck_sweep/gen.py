#!/usr/bin/env python3
"""Tile-configuration sweep for the fused-epilogue convolutions: generates one translation unit per candidate composable_kernel
configuration, builds tools/ck_sweep/libcksweep.so (not part of the product) and, on the GPU (run.py), times every candidate on
the encoder/decoder layer shapes.  Winners are promoted to gdkvm_amd/csrc/conv_ck_t*.hip by hand."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
A4, A8 = "S<4, 64, 1>", "S<8, 32, 1>"
INTRA, INTER = "ck::BlockGemmPipelineScheduler::Intrawave", "ck::BlockGemmPipelineScheduler::Interwave"


def v3(mp, np_, kp, mx, nx, cl, sched, ver, cde="S<1, 32, 1, 8>", blk=256):
    return (f"{blk}, {mp}, {np_}, {kp}, 8, 8, 32, 32, {mx}, {nx}, {cl}, S<1, 0, 2>, S<1, 0, 2>, 2, 8, 8, 0, {cl}, S<1, 0, 2>, "
            f"S<1, 0, 2>, 2, 8, 8, 0, 1, 1, {cde}, 8, {sched}, ck::BlockGemmPipelineVersion::{ver}")


CANDIDATES = {
    "s0_128x64x64_v3": v3(128, 64, 64, 2, 1, A8, INTRA, "v3"),
    "s1_128x64x64_v4": v3(128, 64, 64, 2, 1, A8, INTRA, "v4"),
    "s2_128x64x64_v5": v3(128, 64, 64, 2, 1, A8, INTRA, "v5"),
    "s3_256x64x64_v3": v3(256, 64, 64, 4, 1, A8, INTRA, "v3"),
    "s4_128x64x32_v3": v3(128, 64, 32, 2, 1, A4, INTRA, "v3"),
    "s5_128x64x64_i1": v3(128, 64, 64, 2, 1, A8, INTER, "v1"),
    "s6_64x64x64_v3": v3(64, 64, 64, 1, 1, A8, INTRA, "v3"),
    "s7_256x64x32_v3": v3(256, 64, 32, 4, 1, A4, INTRA, "v3"),
    "s8_128x128x64_v3": v3(128, 128, 64, 2, 2, A8, INTRA, "v3"),
    "s9_256x128x32_v3": v3(256, 128, 32, 4, 2, A4, INTRA, "v3"),
    "s10_128x128x32_v4": v3(128, 128, 32, 2, 2, A4, INTRA, "v4"),
    "s11_256x128x64_v3": v3(256, 128, 64, 4, 2, A8, INTRA, "v3"),
}

TU = '''#include "conv_ck_common.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_grouped_conv_fwd_multiple_abd_xdl_cshuffle_v3.hpp"
namespace gdkvm_ck {{
template <class DsLayout, class DsTypes, class Op>
using Kernel = ck::tensor_operation::device::DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle_V3<2, L::NHWGC, L::GKYXC, DsLayout, L::NHWGK, BF16, BF16,
    F32, F32, DsTypes, BF16, PassThrough, PassThrough, Op, ConvDefault, GemmMNKPadding, {params}>;
}}
extern "C" int {name}(const void* x, const void* w, const float* bias, const void* residual, void* y, int N, int C, int H, int W, int K,
                      int R, int S, int stride, int pad, int relu, void* stream)
{{
    const gdkvm_ck::ConvShape s{{N, C, H, W, K, R, S, stride, pad}};
    return gdkvm_ck::conv_entry<gdkvm_ck::Kernel>(x, w, bias, residual, y, s, relu, static_cast<hipStream_t>(stream));
}}
'''


def main():
    procs, objs = [], []
    for name, params in CANDIDATES.items():
        src = os.path.join(HERE, name + ".hip")
        open(src, "w").write(TU.format(params=params, name=name))
        obj = os.path.join(HERE, name + ".o")
        objs.append((name, obj))
        procs.append((name, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                              "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gdkvm_amd", "csrc"),
                                              "-c", src, "-o", obj], stderr=subprocess.PIPE, text=True)))
        if len(procs) % 6 == 0:
            for _, p in procs[-6:]:
                p.wait()
    good = []
    for (name, p), (_, obj) in zip(procs, objs):
        err = p.communicate()[1]
        if p.returncode == 0:
            good.append(obj)
        else:
            print(f"{name}: does not compile: {[l for l in err.splitlines() if 'error' in l][:2]}")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(HERE, "libcksweep.so")] + good)
    open(os.path.join(HERE, "built.txt"), "w").write("\n".join(os.path.basename(o)[:-2] for o in good))
    print("built", len(good), "of", len(objs))


if __name__ == "__main__":
    sys.exit(main())
