#!/bin/bash
# Turns the outputs of tools/collect_profiles.sh (merged back under gpurun_out/) into the tracked files under profiles/:
#   tools/publish_profiles.sh gpurun_out/r06/final r06
set -e
O=$1; T=${2:-r06}; P=profiles
python3 tools/rocprof_stats_csv.py $O/stats > $P/${T}_bench_kernel_stats.csv
python3 tools/rocprof_stats_csv.py $O/stats_headline > $P/${T}_bench_kernel_stats_headline.csv
python3 tools/rocprof_stats_csv.py $O/stats_headline_overlap0 > $P/${T}_bench_kernel_stats_headline_overlap0.csv
grep '^{' $O/bench_line_headline_overlap0.json | tail -1 > $P/${T}_bench_line_headline_overlap0_under_rocprof.json
grep '^{' $O/bench_line.json | tail -1 > $P/${T}_bench_line_under_rocprof.json
grep '^{' $O/bench_line_headline.json | tail -1 > $P/${T}_bench_line_headline_under_rocprof.json
grep '^{' $O/bench_line_bf16.json | tail -1 > $P/${T}_bench_line_bf16.json
grep '^{' $O/bench_line_plain.json | tail -1 > $P/${T}_bench_line.json
cp $O/layers_f32.txt $P/${T}_layers_f32.txt
cp $O/layers_bf16.txt $P/${T}_layers_bf16.txt
cp $O/layers_f32_mfma_f32.txt $P/${T}_layers_f32_mfma_f32.txt
cp $O/accuracy_modes.txt $P/${T}_accuracy_modes.txt
cp $O/rec_batches.txt $P/${T}_rec_batches.txt
python3 tools/make_pmc_json.py f32 $O/pmc_fetch_f32 $O/pmc_write_f32 $O/pmc_mops_f32 $O/pmc_busy_f32 $T $O/rec_fetch $O/rec_write --labels $O/layers_f32.txt > /dev/null
python3 tools/make_pmc_json.py bf16 $O/pmc_fetch_bf16 $O/pmc_write_bf16 $O/pmc_mops_bf16 $O/pmc_busy_bf16 $T --labels $O/layers_bf16.txt > /dev/null
python3 tools/rocprof_stats_csv.py $O/stats_chain > $P/${T}_pipeline_chain_kernel_stats.csv
python3 tools/rocprof_stats_csv.py $O/stats_fwd > $P/${T}_pipeline_forward_kernel_stats.csv
grep -h "images/s" $O/pipeline_chain.txt $O/pipeline_forward.txt > $P/${T}_pipeline_rates.txt
cp $O/x3w_ab.txt $P/${T}_x3w_ab.txt
cp $O/bf16_block_ab.txt $P/${T}_bf16_block_ab.txt
# the dominant kernels' average duration IN THE SHIPPED SCHEDULE (two streams), from the headline rocprof pass: bench.py's
# roofline.avg_launch_ms_in_schedule
python3 - $P/${T}_bench_kernel_stats_headline.csv <<'PY'
import csv, json, sys
sys.path.insert(0, "tools")
from make_pmc_json import pretty
out = {}
for r in csv.DictReader(open(sys.argv[1])):
    out[pretty(r["Name"])] = {"avg_ms": round(float(r["AverageNs"]) / 1e6, 5), "calls": int(r["Calls"]), "max_ms": round(int(r["MaxNs"]) / 1e6, 5)}
j = json.load(open("profiles/pmc.json"))
j["f32"]["in_schedule"] = {"source": sys.argv[1], "kernels": out}
json.dump(j, open("profiles/pmc.json", "w"), indent=1)
print("in-schedule averages:", {k: v["avg_ms"] for k, v in list(out.items())[:4]})
PY
echo "sources of the passes: $(cat $O/csrc_sha.txt); tree now: $(python3 -c 'import bench; print(bench.csrc_hash())')"
