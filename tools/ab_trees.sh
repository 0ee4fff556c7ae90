#!/bin/bash
# Two builds of the library on ONE box, alternating: the back-to-back step (f32 and bf16) and the per-launch sum.
#   tools/ab_trees.sh <libdir A> <libdir B> [rounds]      (directories under ocr-rs_amd/, e.g. lib lib_old)
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for L in $A $B; do
    echo -n "$L f32 step: "; OCR_AMD_LIB=ocr-rs_amd/$L/libocr_amd.so timeout -k 10 200 python3 tools/bench_overlap.py none 2>&1 | grep -v amdgpu | tail -1
    echo -n "$L bf16 step: "; OCR_AMD_LIB=ocr-rs_amd/$L/libocr_amd.so timeout -k 10 200 python3 tools/bench_overlap.py precision=bf16 2>&1 | grep -v amdgpu | tail -1
  done
done
for L in $A $B; do
  echo "$L per-launch table (one stream, HIP events per launch):"
  OCR_AMD_LIB=ocr-rs_amd/$L/libocr_amd.so timeout -k 10 200 python3 tools/profile_layers.py 32 640 5 0 2>&1 | grep -E "stem|PYR4|PHASE2|k3,s2|BATCHED|total"
done
