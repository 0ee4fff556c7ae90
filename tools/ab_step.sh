#!/bin/bash
# A/B of two builds of the library on ONE box: the back-to-back step of each, alternating.   tools/ab_step.sh <libdir A> <libdir B> [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  echo -n "A $A: "; OCR_AMD_LIB=ocr-rs_amd/$A/libocr_amd.so timeout -k 10 200 python3 tools/bench_overlap.py none 2>&1 | grep -v amdgpu | tail -1
  echo -n "B $B: "; OCR_AMD_LIB=ocr-rs_amd/$B/libocr_amd.so timeout -k 10 200 python3 tools/bench_overlap.py none 2>&1 | grep -v amdgpu | tail -1
done
