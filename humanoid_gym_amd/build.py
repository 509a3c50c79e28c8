"""Builds humanoid_gym_amd/csrc/libhg_sim.so for gfx950 with hipcc (in-tree)."""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libhg_sim.so")
RESOURCES = os.path.join(CSRC, "kernel_resources.json")      # written by build(): registers / scratch per kernel variant
SOURCES = ["hg_sim.hip", "hg_step.hip", "hg_device.h", "hg_math.h",
           os.path.join("..", "..", "include", "hg_sim.h"), os.path.join("..", "..", "include", "hg_model.h")]

# -fno-slp-vectorize: the SLP vectoriser pairs the scalar 3-vector code into v_pk_* ops on
# register pairs, which here costs hundreds of v_mov and ~190 spilled VGPRs per lane.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(os.path.join(CSRC, s)) and os.path.getmtime(os.path.join(CSRC, s)) > t
               for s in SOURCES)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *HIPCC_FLAGS, "-Rpass-analysis=kernel-resource-usage", "hg_sim.hip", "-o", "libhg_sim.so"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    with open(RESOURCES, "w") as f:
        json.dump(parse_resource_remarks(r.stderr), f, indent=1, sort_keys=True)
    return LIB


def parse_resource_remarks(text: str) -> dict:
    """hipcc's kernel-resource-usage remarks -> {"<NMX,MAXC,DEPTH,NSL,WPE>": {"vgprs", "sgprs", "scratch_bytes_per_lane",
    "occupancy_waves_per_simd", "lds_static_bytes"}} for every instantiation of the step kernel (tests assert that
    the variants of the reference's four robots use no scratch)."""
    out = {}
    for block in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = block.split()[0]
        m = re.search(r"hg_step_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        if not m:
            continue

        def g(key):
            k = re.search(key + r": (\d+)", block)
            return int(k.group(1)) if k else None
        out["<" + ",".join(m.groups()) + ">"] = {
            "vgprs": g("VGPRs"), "sgprs": g("TotalSGPRs"), "scratch_bytes_per_lane": g(r"ScratchSize \[bytes/lane\]"),
            "occupancy_waves_per_simd": g(r"Occupancy \[waves/SIMD\]"), "lds_static_bytes": g(r"LDS Size \[bytes/block\]")}
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
