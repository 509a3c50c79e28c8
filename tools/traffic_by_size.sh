#!/bin/bash
# Runs on the GPU box: HBM traffic of one hg_step launch (PMC FETCH_SIZE / WRITE_SIZE, one pass each) at several env
# counts, to separate what scales with the envs from what a launch costs whatever its size (kernel code and model
# tables fetched through each of the eight L2s).
# usage: bash tools/traffic_by_size.sh <tag> <robot> <n_envs> [<n_envs> ...]      -> gpurun_out/traffic_<tag>/
set -u
tag=$1; robot=$2; shift 2
repo=$(pwd)
out=$repo/gpurun_out/traffic_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
    i=0
    for grp in "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1))
        timeout -s KILL 120 rocprofv3 --pmc $grp -d "$out/n$n/pmc$i" -o p --output-format csv -- python3 "$repo/tools/prof_step.py" $robot $n 40 replay > "$out/n$n.pmc$i.log" 2>&1 || { echo "pass $i at $n failed"; tail -5 "$out/n$n.pmc$i.log"; exit 1; }
    done
    python3 "$repo/tools/pmc_summary.py" "$out/n$n" $n > "$out/${robot}_$n.json"
    python3 -c "
import json,sys
d=json.load(open('$out/${robot}_$n.json')); pl=d['per_launch']
print('$robot n_envs=%6d  read %.2f MB  written %.2f MB  total %.2f MB  = %.0f B per env' % ($n, 2*pl['FETCH_SIZE']*1024/1e6, pl['WRITE_SIZE']*1024/1e6, d['hbm_bytes_corrected']/1e6, d['hbm_bytes_corrected']/$n))" | tee -a "$out/summary.txt"
    rm -rf "$out/n$n"
done
