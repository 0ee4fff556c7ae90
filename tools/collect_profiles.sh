#!/bin/bash
# Collects the evidence profiles/ holds for one round, on the GPU box (run through gpurun from the repository root):
#   tools/collect_profiles.sh <outdir under gpurun_out>
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (the JSON line it printed is kept beside it)
# 2. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over one profiled forward and one 65 536-crop recogniser pass
# 3. per-launch layer tables (f32, bf16), the bf16 bench line, accuracy of every engine option, recogniser batches
# Back on the build box: tools/rocprof_stats_csv.py <out>/stats and tools/make_traffic_json.py <out>/pmc_fetch <out>/pmc_write
# r02 <out>/rec_fetch <out>/rec_write turn the rocpd databases into the CSV / JSON files under profiles/.
set -o pipefail
O=$PWD/gpurun_out/${1:-r02/final}
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o t -- python3 bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench.err || exit 1
echo "bench under rocprof done"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o t -- python3 tools/profile_layers.py 32 640 1 > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o t -- python3 tools/profile_layers.py 32 640 1 > /dev/null 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE -d $O/rec_fetch -o t -- python3 tools/profile_rec.py 65536 > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/rec_write -o t -- python3 tools/profile_rec.py 65536 > /dev/null 2>&1 || exit 1
echo "pmc passes done"
python3 tools/profile_layers.py 32 640 5 > $O/layers_f32.txt 2>/dev/null || exit 1
python3 tools/profile_layers.py 32 640 5 0 "precision=bf16" > $O/layers_bf16.txt 2>/dev/null || exit 1
python3 bench.py --steps 20 --warmup 3 --dtype bf16 > $O/bench_line_bf16.json 2>/dev/null || exit 1
python3 tools/accuracy_report.py > $O/accuracy_modes.txt 2>/dev/null || exit 1
python3 tools/bench_rec.py > $O/rec_batches.txt 2>/dev/null || exit 1
python3 bench.py --steps 20 --warmup 3 > $O/bench_line_plain.json 2>/dev/null || exit 1
echo "all done"
