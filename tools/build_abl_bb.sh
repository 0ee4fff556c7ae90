#!/bin/bash
# Compile-time ablations of basic_block_bf16_c64.hip (BB_ABL bits: 1 no fragment reads, 2 no MFMAs, 4 no patch DMA, 8 no stores of the
# intermediate, 16 no conv2 epilogue, 32 conv1 waves idle, 64 conv2 waves idle): builds ocr-rs_amd/lib_bb<N> from the objects of
# ocr-rs_amd/lib with basic_block_bf16_c64.o recompiled.
#   tools/build_abl_bb.sh <N> [<N> ...]; then on the GPU box: for every N  OCR_AMD_LIB=ocr-rs_amd/lib_bb<N>/libocr_amd.so python3 tools/bench_bf16_block.py abl
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for N in "$@"; do
  D=$ROOT/ocr-rs_amd/lib_bb$N
  rm -rf "$D" && mkdir -p "$D/obj" && cp "$ROOT"/ocr-rs_amd/lib/obj/*.o "$D/obj/"
  (cd "$ROOT/ocr-rs_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DBB_ABL=$N $BB_EXTRA -c basic_block_bf16_c64.hip -o "$D/obj/basic_block_bf16_c64.o" 2>/dev/null)
  OBJS=$(ls "$D"/obj/*.o | grep -v test_hooks)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$D/libocr_amd_test.so" "$D/obj/test_hooks.o" $OBJS -ldl
  cp "$D/libocr_amd_test.so" "$D/libocr_amd.so"
  rm -rf "$D/obj"
  echo "built lib_bb$N"
done
