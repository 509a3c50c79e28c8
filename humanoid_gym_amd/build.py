"""Builds humanoid_gym_amd/csrc/libhg_sim.so for gfx950 with hipcc (in-tree)."""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libhg_sim.so")
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
    cmd = [hipcc, *HIPCC_FLAGS, "hg_sim.hip", "-o", "libhg_sim.so"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
