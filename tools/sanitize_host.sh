#!/bin/bash
# AddressSanitizer + UBSan over the HOST code of the product (CPU build only - no GPU sanitizers on this pool):
#   1. post-processing geometry (postproc_geom.cpp: bit-image contour tracer, Douglas-Peucker, min-area rectangle, offset +
#      exact-rational union) through oracle/postproc_cpu.cpp: the reference's known answers, then text-like / dense pages,
#      noise, stripes, checkerboards, full maps and random rectangles at several sizes, 1 and 3 threads;
#   2. the VarStore reader (varstore.cpp: zip + pickle subset) on 20 000 mutations of a valid archive (bit flips, random
#      bytes, truncations, 0xff runs, splices): every outcome must be a blob or an ocr::Error.
# Any sanitizer report fails the run.  Usage: tools/sanitize_host.sh   (about two minutes)
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
trap 'rm -rf "$T"; make -s -B -C oracle libpostproc_cpu.so > /dev/null' EXIT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
g++ $SAN -std=c++17 -fPIC -ffp-contract=off -pthread -shared -o oracle/libpostproc_cpu.so oracle/postproc_cpu.cpp ocr-rs_amd/csrc/postproc_geom.cpp
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
LD_PRELOAD="$PRE" python3 -m pytest tests/test_oracle_postproc.py -x -q 2>&1 | tee $T/a.log | tail -2
LD_PRELOAD="$PRE" python3 tools/fuzz/postproc_fuzz.py 2>&1 | tee $T/b.log | tail -2
g++ $SAN -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iocr-rs_amd/csrc -Iinclude tools/fuzz/varstore_fuzz.cpp ocr-rs_amd/csrc/varstore.cpp -o $T/varstore_fuzz
ASAN_OPTIONS=detect_leaks=1:allocator_may_return_null=1 $T/varstore_fuzz tests/golden/varstore_small.ot 20000 $T 2>&1 | tee $T/c.log | tail -2
if grep -q "runtime error\|AddressSanitizer\|LeakSanitizer" $T/a.log $T/b.log $T/c.log; then echo "SANITIZER REPORTS"; exit 1; fi
echo "host sanitizer run clean"
