"""One full resident round per launch: n_envs = CUs x k with at most k waves per CU (HG_WAVES_PER_CU=k), k = 1 .. the
variant's residency.  Round time and throughput per CU against the residency: how much of the VALU pipe does a round of k
waves per CU use, and what would more resident waves buy?  usage (GPU box): python tools/residency_sweep.py [robot]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
robot = sys.argv[1] if len(sys.argv) > 1 else "nao_hands"
code = """
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tools'))
import numpy as np, torch, bench
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd._capi import HgSim
from humanoid_gym_amd.workloads import nao_replay_joints
k = int(os.environ['HG_WAVES_PER_CU']); n_envs = 256 * k
model, cfg = load_robot(%r); cfg.store_torques = False
rows, sel = bench.synthetic_replay(model)
if %r == 'nao_hands':
    rows, sel = bench.with_mimic_columns(model, rows, sel, nao_replay_joints())
sim = HgSim(model, cfg, n_envs, 'cuda:0')
rows_d = torch.tensor(rows, device='cuda:0'); T = rows.shape[0]
phase = torch.arange(n_envs, device='cuda:0') %% T
acts = torch.stack([rows_d[(t + phase) %% T] for t in range(T)]).contiguous()
for t in range(60): sim.step(acts[t %% T], joint_sel=sel)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(100): sim.step(acts[t %% T], joint_sel=sel)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 100
print(f'{k:2d} waves per CU: {n_envs:5d} envs, {ms * 1e3:7.1f} us per launch, {n_envs / ms / 1e3:6.2f} M env-steps/s, {ms * 1e3 / k:6.1f} us per wave and CU')
""" % (ROOT, ROOT, robot, robot)
for k in range(1, 17):
    env = dict(os.environ, HG_WAVES_PER_CU=str(k))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print((out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
