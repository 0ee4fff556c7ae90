#!/bin/bash
# Collects the evidence profiles/ holds for one round, on the GPU box (run through gpurun from the repository root):
#   tools/collect_profiles.sh <outdir under gpurun_out>
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (the JSON line it printed is kept beside it)
# 2. PMC passes over one profiled forward, every counter set in its OWN run (no tracing besides the kernel trace): FETCH_SIZE,
#    WRITE_SIZE, MFMA op counters, MFMA-busy cycles - for the f32 and the bf16 precision - and FETCH / WRITE over one
#    65 536-crop recogniser pass
# 3. per-launch layer tables (f32, f32 with mfma=f32, bf16), the bf16 bench line, accuracy of every engine option, recogniser batches
# Back on the build box: tools/rocprof_stats_csv.py <out>/stats and tools/make_pmc_json.py turn the outputs into the CSV / JSON
# files under profiles/.
set -o pipefail
O=$PWD/gpurun_out/${1:-r06/final}
mkdir -p $O
export TMPDIR=/tmp
python3 -c "import bench; print(bench.csrc_hash())" > $O/csrc_sha.txt   # the kernel sources these passes ran (stamps profiles/pmc.json)
rocprofv3 --kernel-trace --stats -d $O/stats -o t -- python3 bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench.err || exit 1
# the same with the headline loop and the roofline passes only: every detector launch is a batch-32 forward, so the
# average duration per kernel is the one bench.py's HIP events report (the full command also runs 8-frame pieces for the
# host-memory rates and other weights, which share kernel names)
rocprofv3 --kernel-trace --stats -d $O/stats_headline -o t -- python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_line_headline.json 2> /dev/null || exit 1
# ... and with the engine's side stream off (overlap=0): every launch of the dominant kernel runs alone, as in bench.py's own per-launch
# HIP-event passes (ocr_det_forward_profile is always one stream) - the run whose rocprof average the roofline's avg_launch_ms must match
rocprofv3 --kernel-trace --stats -d $O/stats_headline_overlap0 -o t -- python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --det-options overlap=0 > $O/bench_line_headline_overlap0.json 2> /dev/null || exit 1
echo "bench under rocprof done"
for P in f32 bf16; do
  OPT=""; [ $P = bf16 ] && OPT="precision=bf16"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$P -o t -- python3 tools/profile_layers.py 32 640 1 0 "$OPT" > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$P -o t -- python3 tools/profile_layers.py 32 640 1 0 "$OPT" > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_mops_$P -o t -- python3 tools/profile_layers.py 32 640 1 0 "$OPT" > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_busy_$P -o t -- python3 tools/profile_layers.py 32 640 1 0 "$OPT" > /dev/null 2>&1 || exit 1
  echo "pmc passes $P done"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/rec_fetch -o t -- python3 tools/profile_rec.py 65536 > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/rec_write -o t -- python3 tools/profile_rec.py 65536 > /dev/null 2>&1 || exit 1
echo "recogniser pmc passes done"
python3 tools/profile_layers.py 32 640 5 > $O/layers_f32.txt 2>/dev/null || exit 1
python3 tools/profile_layers.py 32 640 5 0 "mfma=f32" > $O/layers_f32_mfma_f32.txt 2>/dev/null || exit 1
python3 tools/profile_layers.py 32 640 5 0 "precision=bf16" > $O/layers_bf16.txt 2>/dev/null || exit 1
python3 bench.py --steps 20 --warmup 3 --dtype bf16 > $O/bench_line_bf16.json 2>/dev/null || exit 1
python3 tools/accuracy_report.py > $O/accuracy_modes.txt 2>/dev/null || exit 1
python3 tools/bench_rec.py > $O/rec_batches.txt 2>/dev/null || exit 1
# the polygon chain beside the forward: per-kernel durations of pipelined detect calls (dense pages, chain on the device) and of forwards alone
rocprofv3 --kernel-trace --stats -d $O/stats_chain -o t -- python3 tools/pipeline_trace.py chain > $O/pipeline_chain.txt 2>/dev/null || exit 1
rocprofv3 --kernel-trace --stats -d $O/stats_fwd -o t -- python3 tools/pipeline_trace.py forward > $O/pipeline_forward.txt 2>/dev/null || exit 1
# the 256 x 128 persistent split-bf16 form against the 128-wide tiles, launch shape by launch shape
python3 tools/bench_x3w.py > $O/x3w_ab.txt 2>/dev/null || exit 1
# bf16 precision: a BasicBlock of layer1 as one launch against its two launches
python3 tools/bench_bf16_block.py > $O/bf16_block_ab.txt 2>/dev/null || exit 1
echo "chain / wide-form passes done"
python3 bench.py --steps 20 --warmup 3 > $O/bench_line_plain.json 2>/dev/null || exit 1
echo "all done"
