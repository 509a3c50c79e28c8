#!/bin/bash
# VALU/SALU/LDS instruction counts per wave for a few configurations (GPU box).
set -u
repo=$(pwd); out=$repo/gpurun_out/pmcv; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for cfgargs in "" "solver_iters=0" "enable_ground=0" "enable_ground=0 solver_iters=0"; do
    i=$((i+1))
    timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$out/v$i" -o p --output-format csv -- python3 "$repo/tools/prof_step.py" nao 4096 6 $cfgargs > "$out/v$i.log" 2>&1 || { echo "pass $i failed"; exit 1; }
    echo "== $cfgargs"
    python3 "$repo/tools/pmc_summary.py" "$out/v$i" 4096 | grep -A8 per_wave | grep SQ_
    rm -rf "$out/v$i"
done
