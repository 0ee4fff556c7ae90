#!/bin/bash
# Compile-time ablations of conv_x3w.hip (X3W_ABL bits: 1 no A DMA, 2 no B DMA, 4 no split, 8 no MFMA, 32 no LDS fragment reads, 64 no mid-step barrier,
# 128 no epilogue, 512 no wait for the DMA): builds ocr-rs_amd/lib_x3w<N> from the objects of ocr-rs_amd/lib with conv_x3w.o recompiled.
#   tools/build_abl_x3w.sh <N> [<N> ...]; then on the GPU box: for every N  OCR_AMD_LIB=ocr-rs_amd/lib_x3w<N>/libocr_amd.so python3 tools/bench_x3w.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for N in "$@"; do
  D=$ROOT/ocr-rs_amd/lib_x3w$N
  rm -rf "$D" && mkdir -p "$D/obj" && cp "$ROOT"/ocr-rs_amd/lib/obj/*.o "$D/obj/"
  (cd "$ROOT/ocr-rs_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DX3W_ABL=$N -c conv_x3w.hip -o "$D/obj/conv_x3w.o" 2>/dev/null)
  OBJS=$(ls "$D"/obj/*.o | grep -v test_hooks)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$D/libocr_amd_test.so" "$D/obj/test_hooks.o" $OBJS -ldl
  cp "$D/libocr_amd_test.so" "$D/libocr_amd.so"
  echo "built lib_x3w$N"
done
