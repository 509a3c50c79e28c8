#!/bin/bash
# Per-variant register / scratch / code-size report of the step kernel (hipcc remarks + ISA line count).
# usage: bash tools/kernel_resources.sh [extra hipcc flags]      (runs on CPU: hipcc cross-compiles)
cd "$(dirname "$0")/../humanoid_gym_amd/csrc" || exit 1
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC hg_sim.hip -o "$tmp/lib.so" \
    -Rpass-analysis=kernel-resource-usage -save-temps=obj "$@" 2> "$tmp/remarks.txt"
python3 - "$tmp" <<'PY'
import re, sys, glob
tmp = sys.argv[1]
txt = open(tmp + "/remarks.txt").read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
asm = ""
for f in glob.glob(tmp + "/*gfx950*.s"):
    asm = open(f).read()
for b in blocks:
    name = b.split()[0]
    if "hg_step_kernel" not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    m = re.search(r"hg_step_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
    body = asm.split(name + ":", 1)[1].split(".end_amdhsa_kernel", 1)[0] if name + ":" in asm else ""
    n_inst = sum(1 for ln in body.splitlines() if ln.startswith("\t") and not ln.startswith("\t.") and not ln.strip().startswith(";"))
    n_valu = sum(1 for ln in body.splitlines() if ln.startswith("\tv_"))
    scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"<{','.join(m.groups())}>  VGPRs {g('VGPRs'):>3}  AGPRs {g('AGPRs'):>3}  SGPRs {g('TotalSGPRs'):>3}  scratch {scr:>4} B/lane  "
          f"occupancy {occ}  static insts {n_inst} (VALU {n_valu})")
PY
rm -rf "$tmp"
