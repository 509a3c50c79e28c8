#!/bin/bash
# Runs on the GPU box: kernel-trace stats of the bench command, then PMC passes (one counter group per run).
# usage: bash tools/collect_profiles.sh <tag> [robot]      -> gpurun_out/prof_<tag>/      (robot: nao_hands = the headline, or nao)
set -u
tag=${1:-run}
robot=${2:-nao_hands}
repo=$(pwd)
out=$repo/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$out/stats" -o bench --output-format csv -- python3 "$repo/bench.py" --robot $robot --no-cpu-baseline --stream-groups 0 > "$out/bench_under_rocprof.log" 2>&1 || { echo "stats pass failed"; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout -s KILL 90 rocprofv3 --pmc $grp -d "$out/pmc$i" -o p --output-format csv -- python3 "$repo/tools/prof_step.py" $robot 4096 70 replay > "$out/pmc$i.log" 2>&1 || { echo "pmc pass $i failed"; tail -5 "$out/pmc$i.log"; exit 1; }
    echo "pmc pass $i ok"
done
python3 "$repo/tools/pmc_summary.py" "$out" 4096 > "$out/pmc_summary.json"
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
cat "$out/pmc_summary.json"
head -5 "$out/kernel_stats.csv"
tail -1 "$out/bench_under_rocprof.log" | cut -c1-300
# keep only the summaries (the raw traces are large)
rm -rf "$out/stats" "$out"/pmc[0-9]
