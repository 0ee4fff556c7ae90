#!/bin/bash
# per-launch times of the split-bf16 convs under compile-time ablations: libraries ocr-rs_amd/lib_abl<N> built with -DIGEMM_ABL=<N> (conv_igemm.hip)
for n in ${ABLS:-0 1 2 4 8 32 40 64 128 512}; do
  [ $n = 0 ] || [ -d ocr-rs_amd/lib_abl$n ] || continue
  L=ocr-rs_amd/lib_abl$n; [ $n = 0 ] && L=ocr-rs_amd/lib
  echo "== ABL $n"
  OCR_AMD_LIB=$L/libocr_amd.so timeout -k 10 120 python3 tools/profile_layers.py 32 640 4 0 2>&1 | grep -E "conv_igemm"
done
