"""Builds humanoid_gym_amd/csrc/libhg_sim.so for gfx950 with hipcc (in-tree).

The step kernel is a template with one instantiation per robot shape (csrc/hg_variants.h).  Each instantiation is its
own translation unit (csrc/hg_step_variant.hip -DHG_VI=<index>), so they compile in parallel; hg_sim.hip (C-ABI, helper
kernels) is one more, and the objects are linked into the shared library.  Objects live in csrc/_obj/ (git- and
gpurun-ignored); a variant object is rebuilt when any source is newer than it.
"""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libhg_sim.so")
RESOURCES = os.path.join(CSRC, "kernel_resources.json")      # written by build(): registers / scratch per kernel variant
SOURCES = ["hg_sim.hip", "hg_step.hip", "hg_step_variant.hip", "hg_step_launch.h", "hg_variants.h", "hg_device.h", "hg_math.h",
           os.path.join("..", "..", "include", "hg_sim.h"), os.path.join("..", "..", "include", "hg_model.h")]

# -fno-slp-vectorize: the SLP vectoriser pairs the scalar 3-vector code into v_pk_* ops on
# register pairs, which here costs hundreds of v_mov and ~190 spilled VGPRs per lane.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC"]
COMPILE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-DHG_SPLIT_TU"]


def variant_rows():
    """[(index, nmx, maxc, depth, nsl, wpe)] from the X rows of csrc/hg_variants.h."""
    txt = open(os.path.join(CSRC, "hg_variants.h")).read()
    return [tuple(int(x) for x in m.groups())
            for m in re.finditer(r"X\((\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", txt)]


def _newest_source() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, s)) for s in SOURCES if os.path.exists(os.path.join(CSRC, s)))


def needs_build(lib: str = LIB) -> bool:
    return not os.path.exists(lib) or _newest_source() > os.path.getmtime(lib)


def build(force: bool = False, verbose: bool = False, lib: str = LIB, extra_flags=(), obj_dir: str = "_obj",
          jobs: int | None = None) -> str:
    """Compiles what is out of date and links `lib`.  `extra_flags` (e.g. -DHG_PHASE_TIMING for tools/phase_ticks.py)
    go to every compile; give such a build its own `obj_dir`."""
    if not force and not needs_build(lib):
        return lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    odir = os.path.join(CSRC, obj_dir)
    os.makedirs(odir, exist_ok=True)
    newest = _newest_source()
    units = [("hg_sim", "hg_sim.hip", [])] + [(f"hg_step_v{r[0]}", "hg_step_variant.hip", [f"-DHG_VI={r[0]}"]) for r in variant_rows()]
    remarks_path = lambda name: os.path.join(odir, name + ".remarks.txt")

    def compile_unit(u):
        name, src, defs = u
        obj = os.path.join(odir, name + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest and os.path.exists(remarks_path(name)):
            return name, 0, ""
        cmd = [hipcc, *COMPILE_FLAGS, *extra_flags, *defs, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
        if r.returncode == 0:
            with open(remarks_path(name), "w") as f:
                f.write(r.stderr)
        return name, r.returncode, r.stderr

    jobs = jobs or int(os.environ.get("MAX_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        results = list(pool.map(compile_unit, units))
    for name, rc, err in results:
        if rc != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n" + err[-4000:])
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[os.path.join(odir, u[0] + ".o") for u in units], "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    if lib == LIB:
        remarks = "".join(open(remarks_path(u[0])).read() for u in units)
        with open(RESOURCES, "w") as f:
            json.dump(parse_resource_remarks(remarks), f, indent=1, sort_keys=True)
    return lib


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
