#!/bin/bash
# Builds tests/host_san/_build/host_san: the host halves of four product translation units + the driver; host code under ASan + UBSan, device code compiled as usual (-fno-gpu-sanitize) and never run.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(dirname "$(dirname "$HERE")")"; CSRC="$ROOT/gdkvm_amd/csrc"
mkdir -p "$HERE/_build"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-gpu-sanitize -x hip -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -O1 -g -std=c++17 \
    -I"$ROOT/include" -I"$CSRC" "$HERE/host_san_driver.cpp" "$CSRC/gdkvm_api.hip" "$CSRC/gdr_segmented.hip" "$CSRC/gdr_normalizer.hip" "$CSRC/gdr_step.hip" \
    -o "$HERE/_build/host_san"
