#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats + PMC passes (one counter group per run) of the step kernel for the
# BASELINE configs other than the NAO headline, at their per-GPU sizes, random targets, ground + link-link contacts.
# usage: bash tools/collect_variant_profiles.sh <tag>   -> gpurun_out/vprof_<tag>/<robot>/{kernel_stats.csv,pmc_summary.json}
set -u
tag=${1:-run}
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
while read -r robot envs sub launches; do
    out=$repo/gpurun_out/vprof_$tag/$robot
    mkdir -p "$out"
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/stats" -o k --output-format csv -- python3 "$repo/tools/prof_step.py" $robot $envs $launches $sub random > "$out/stats.log" 2>&1 || { echo "$robot: stats pass failed"; tail -3 "$out/stats.log"; exit 1; }
    i=0
    for grp in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" \
               "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT"; do
        i=$((i+1))
        timeout -s KILL 120 rocprofv3 --pmc $grp -d "$out/pmc$i" -o p --output-format csv -- python3 "$repo/tools/prof_step.py" $robot $envs 12 $sub random > "$out/pmc$i.log" 2>&1 || { echo "$robot: pmc pass $i failed"; tail -3 "$out/pmc$i.log"; exit 1; }
    done
    python3 "$repo/tools/pmc_summary.py" "$out" $envs > "$out/pmc_summary.json"
    find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
    rm -rf "$out/stats" "$out"/pmc[0-9]
    echo "== $robot $envs envs x $sub substeps"; head -2 "$out/kernel_stats.csv" | cut -c1-200
    python3 - "$out/pmc_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); pl, pw = d["per_launch"], d["per_wave"]
clk = pl.get("GRBM_GUI_ACTIVE", 0) / 8            # per XCD
print(f"  VGPRs {d.get('vgprs')} scratch {d.get('scratch_bytes_per_lane')} B/lane; VALU/wave {pw.get('SQ_INSTS_VALU', 0):.0f}; "
      f"HBM {d.get('hbm_bytes_corrected', 0) / 1e6:.2f} MB/launch; GPU-active clocks/XCD {clk:.0f}")
PY
done <<'CFG'
pepper 16384 1 40
romeo 8192 1 40
dancer 65536 4 24
nao_hands 4096 1 40
romeo_hands 8192 1 24
CFG
