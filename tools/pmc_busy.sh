#!/bin/bash
# How busy is the VALU?  One PMC pass (GPU box).  usage: bash tools/pmc_busy.sh [prof_step args...]
set -u
repo=$(pwd); out=$repo/gpurun_out/pmcb; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU -d "$out/b" -o p --output-format csv -- python3 "$repo/tools/prof_step.py" nao 4096 6 "$@" > "$out/b.log" 2>&1 || { echo "pass failed"; tail -3 "$out/b.log"; exit 1; }
python3 "$repo/tools/pmc_summary.py" "$out/b" 4096 | grep -B1 -A12 per_launch
rm -rf "$out/b"
