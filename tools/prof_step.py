"""Minimal driver for rocprofv3 passes: N launches of hg_step on one robot/config.
usage: python tools/prof_step.py <robot> <n_envs> <launches> [n_substeps] [random] [cfg_key=value ...]
`random`: new uniform targets inside the joint limits every 10 launches (random_control_*.py) instead of holding the reset targets.
`replay` (nao, nao_hands): the bench workload -- the h5-layout trajectory replay (+ the fingers' mimic columns), every env at its own frame."""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd._capi import HgSim
from humanoid_gym_amd.workloads import random_actions
robot = sys.argv[1] if len(sys.argv) > 1 else "nao"
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rest = sys.argv[4:]
sub = int(rest.pop(0)) if rest and rest[0].isdigit() else 1
rand = bool(rest and rest[0] == "random")
replay = bool(rest and rest[0] == "replay")
if rand or replay:
    rest.pop(0)
model, cfg = load_robot(robot)
if replay:
    cfg.store_torques = False      # as bench.py / NaoEnv: nothing on that path reads the motor torques,
    cfg.store_contacts = False     # the contact lists
for kv in rest:
    k, v = kv.split("=")
    setattr(cfg, k, type(getattr(cfg, k))(ast.literal_eval(v)))
sim = HgSim(model, cfg, n_envs, "cuda:0")
if replay:
    sim.obs_qd = None              # or the joint velocities
rng = np.random.default_rng(4)
acts = None
if replay:
    import bench
    from humanoid_gym_amd.workloads import nao_replay_joints
    rows, sel = bench.synthetic_replay(model)
    if robot == "nao_hands":
        rows, sel = bench.with_mimic_columns(model, rows, sel, nao_replay_joints())
    rows_d = torch.tensor(rows, device="cuda:0")
    phase = torch.arange(n_envs, device="cuda:0") % bench.T_TRAJ
    for t in range(reps):
        sim.step(rows_d[(t + phase) % bench.T_TRAJ].contiguous(), joint_sel=sel, n_substeps=sub)
    reps = 0
for t in range(reps):
    if rand and t % 10 == 0:
        a = rng.uniform(-1, 1, (n_envs, 18)).astype(np.float32) if robot == "dancer" else random_actions(model, rng, n_envs)
        acts = torch.tensor(a, device="cuda:0")
    sim.step(acts, n_substeps=sub)
torch.cuda.synchronize()
print("done", robot, n_envs, reps, sub, "random" if rand else "hold")
