"""Minimal driver for rocprofv3 passes: N launches of hg_step on one robot/config."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd._capi import HgSim
robot = sys.argv[1] if len(sys.argv) > 1 else "nao"
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
model, cfg = load_robot(robot)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    setattr(cfg, k, type(getattr(cfg, k))(eval(v)))
sim = HgSim(model, cfg, n_envs, "cuda:0")
for _ in range(reps):
    sim.step(None)
torch.cuda.synchronize()
print("done", robot, n_envs, reps)
