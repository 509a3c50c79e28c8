"""Where does the time of one hg_step launch go?  Builds a diagnostics copy of the library with
-DHG_PHASE_TIMING (the step kernel stamps the shader clock, s_memtime, at every phase boundary; the product
library carries no such code), runs the bench workload and prints per-phase durations of a wave.

    python tools/phase_ticks.py build      # here, on the CPU (hipcc cross-compiles)
    python tools/phase_ticks.py [robot] [n_envs]    # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "humanoid_gym_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "libhg_sim_ticks.so")
PHASES = ["prologue: topology tables, targets", "state + link constants, sin/cos, base pose", "kinematics level sweep",
          "contact detection", "ABA leaves -> root", "base 6x6 solve, ABA root -> leaves",
          "row set-up: contact Jacobians, motor / limit rows", "impulse responses (rows of A)", "PGS sweeps",
          "impulses -> joints, final response", "integration, state store", "epilogue: guard, observations, flags"]

if len(sys.argv) > 1 and sys.argv[1] == "build":
    from humanoid_gym_amd.build import build
    print(build(force="--force" in sys.argv, verbose=True, lib=LIB, extra_flags=["-DHG_PHASE_TIMING"], obj_dir="_obj_ticks"))
    sys.exit(0)

import numpy as np
import torch
import bench
from humanoid_gym_amd import _capi
from humanoid_gym_amd.robots import load_robot

import ast
robot = sys.argv[1] if len(sys.argv) > 1 else "nao"
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
overrides = dict(a.split("=", 1) for a in sys.argv[3:])      # SimConfig overrides, e.g. max_contacts=6
L = _capi.load_library(LIB)
_capi._lib = L
L.hg_debug_phase_ticks.restype = C.c_int
L.hg_debug_phase_ticks.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda", 0)
model, cfg = load_robot(robot)
for k, v in overrides.items():
    setattr(cfg, k, type(getattr(cfg, k))(ast.literal_eval(v)))
if robot in ("nao", "nao_hands"):          # the bench workload: h5-layout replay (+ the fingers' mimic columns)
    rows, sel = bench.synthetic_replay(model)
    if robot == "nao_hands":
        from humanoid_gym_amd.workloads import nao_replay_joints
        rows, sel = bench.with_mimic_columns(model, rows, sel, nao_replay_joints())
else:
    sel = np.asarray(model.act_joint, np.int32)
    rows = np.random.default_rng(2).uniform(model.lower[sel], model.upper[sel], (bench.T_TRAJ, len(sel))).astype(np.float32)
sim = _capi.HgSim(model, cfg, n_envs, dev)
rows_d = torch.tensor(rows, device=dev)
T = rows.shape[0]
phase = torch.arange(n_envs, device=dev) % T
acts = torch.stack([rows_d[(t + phase) % T] for t in range(T)]).contiguous()
for t in range(60):
    sim.step(acts[t % T], joint_sel=sel)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(60, 100):
    sim.step(acts[t % T], joint_sel=sel)
e1.record(); torch.cuda.synchronize()
print(f"{robot} {n_envs} envs {overrides}: {e0.elapsed_time(e1) / 40:.4f} ms per step (diagnostics build), variant {os.environ.get('HG_VARIANT', 'auto')}")
ne = min(n_envs, 4096)
buf = np.zeros((ne, 24), np.uint64)
assert L.hg_debug_phase_ticks(buf.ctypes.data_as(C.c_void_p), ne) == 24
tk = buf[:, :13].astype(np.int64)
d = np.diff(tk, axis=1)                       # [env, 12] clocks per phase
total = tk[:, 12] - tk[:, 0]
hw, xcc = buf[:, 13].astype(np.int64), buf[:, 14].astype(np.int64) & 0xf
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
# every XCD has its own clock: starts and ends are taken relative to the first start on the same XCD
start, end = np.zeros(ne), np.zeros(ne)
for x in np.unique(xcc):
    m = xcc == x
    t0 = tk[m, 0].min()
    start[m], end[m] = tk[m, 0] - t0, tk[m, 12] - t0
simd_key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
uniq, per_simd = np.unique(simd_key, return_counts=True)
cu_key = simd_key // 4
_, per_cu = np.unique(cu_key, return_counts=True)
kc = 1e-3
print(f"{robot}, {n_envs} envs, one launch; shader clocks (s_memtime), in thousands")
print(f"XCDs seen {len(np.unique(xcc))}, CUs seen {len(per_cu)}, SIMDs seen {len(uniq)}; waves per SIMD: "
      + ", ".join(f"{k}: {int((per_simd == k).sum())} SIMDs" for k in np.unique(per_simd))
      + "; waves per CU: " + ", ".join(f"{k}: {int((per_cu == k).sum())}" for k in np.unique(per_cu)))
print(f"wave start after the XCD's first: mean {start.mean() * kc:.1f}k, max {start.max() * kc:.1f}k; "
      f"last end {end.max() * kc:.1f}k; lifetime mean {total.mean() * kc:.1f}k, min {total.min() * kc:.1f}k, max {total.max() * kc:.1f}k")
cnt_of = dict(zip(uniq.tolist(), per_simd.tolist()))
mates = np.array([cnt_of[k] for k in simd_key.tolist()])
for k in np.unique(mates):
    print(f"  waves on SIMDs holding {k}: lifetime mean {total[mates == k].mean() * kc:.1f}k, PGS mean {d[mates == k, 8].mean() * kc:.1f}k")
wid = hw & 0xf
for k in np.unique(wid):
    m = wid == k
    print(f"  wave slot {k}: {int(m.sum())} waves, lifetime mean {total[m].mean() * kc:.1f}k, PGS mean {d[m, 8].mean() * kc:.1f}k, "
          f"before PGS {(tk[m, 8] - tk[m, 0]).mean() * kc:.1f}k")
if buf[:, 15].any():
    ov = (buf[:, 15].astype(np.int64) - tk[:, 7]) * kc
    print(f"  of the impulse responses: overflow rows' path-only response {ov.mean():.2f}k (min {ov.min():.2f}k, max {ov.max():.2f}k)")
if buf[:, 16].any():
    t16, t17 = buf[:, 16].astype(np.int64), buf[:, 17].astype(np.int64)
    first = np.where(buf[:, 15] != 0, buf[:, 15].astype(np.int64), tk[:, 7])
    print(f"  of the response of the rows that have a lane: own-path walk {(t16 - first).mean() * kc:.2f}k, base {(t17 - t16).mean() * kc:.2f}k, "
          f"descent over the links {(tk[:, 8] - t17).mean() * kc:.2f}k")
print(f"{'phase':58s} {'mean kclk':>9s} {'min':>7s} {'max':>7s} {'share':>6s}")
for k, name in enumerate(PHASES):
    c = d[:, k] * kc
    print(f"{name:58s} {c.mean():9.2f} {c.min():7.2f} {c.max():7.2f} {100 * c.mean() / (total.mean() * kc):5.1f}%")
sim.close()
