#!/bin/bash
# Differential fuzzing campaign on the GPU box (tools/fuzz_robots.py, tools/fuzz_trees.py); outliers and failures are
# dumped for tools/fuzz_replay.py.  usage: bash tools/fuzz_campaign.sh <tag> [robot seeds] [tree seeds]
tag=${1:-run}; nrs=${2:-20}; nts=${3:-4}
out=gpurun_out/fuzz_$tag
mkdir -p $out
export FUZZ_DUMP=$out/dump
for seed in $(seq 500 $((500 + nrs - 1))); do
    timeout -k 10 120 python tools/fuzz_robots.py 80 $seed >> $out/robots.log 2>&1 || echo "seed $seed: exit $?" >> $out/robots.log
    tail -1 $out/robots.log | cut -c1-200
done
for seed in $(seq 61 $((61 + nts - 1))); do
    timeout -k 10 300 python tools/fuzz_trees.py 500 $seed >> $out/trees.log 2>&1 || echo "seed $seed: exit $?" >> $out/trees.log
    tail -1 $out/trees.log | cut -c1-200
done
grep -h "explained outlier\|FAILED\|fuzz robots:\|fuzz trees:\|exit" $out/robots.log $out/trees.log > $out/summary.txt
