#!/bin/bash
# More seeds of the same campaign (tools/fuzz_campaign.sh covers robot seeds 500.. and tree seeds 61..): robot seeds 520-579,
# tree seeds 65-76 (or <robot seed 0> <tree seed 0>: 60 robot seeds and 12 tree seeds from there).
# usage (GPU box): bash tools/fuzz_campaign_more.sh <tag> [robot seed 0] [tree seed 0]
tag=${1:-more}; r0=${2:-520}; t0=${3:-65}
out=gpurun_out/fuzz_$tag
mkdir -p $out
export FUZZ_DUMP=$out/dump
for seed in $(seq $r0 $((r0 + 59))); do
    timeout -k 10 120 python tools/fuzz_robots.py 80 $seed >> $out/robots.log 2>&1 || echo "seed $seed: exit $?" >> $out/robots.log
    echo "robots $seed: $(tail -1 $out/robots.log | cut -c1-120)"
done
for seed in $(seq $t0 $((t0 + 11))); do
    timeout -k 10 300 python tools/fuzz_trees.py 500 $seed >> $out/trees.log 2>&1 || echo "seed $seed: exit $?" >> $out/trees.log
    echo "trees $seed: $(tail -1 $out/trees.log | cut -c1-120)"
done
grep -h "explained outlier\|FAILED\|fuzz robots:\|fuzz trees:\|exit" $out/robots.log $out/trees.log > $out/summary.txt
