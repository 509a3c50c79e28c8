"""Times hg_step under config ablations (no rebuild): which phase dominates?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd._capi import HgSim

def run(name, n_envs, reps=50, **kw):
    model, cfg = load_robot(name)
    for k, v in kw.items():
        setattr(cfg, k, v)
    sim = HgSim(model, cfg, n_envs, "cuda:0")
    for _ in range(30):
        sim.step(None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sim.step(None)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    _, counts, _ = sim.contact_pairs()
    print(f"{name:7s} envs={n_envs:6d} {kw} -> {ms*1e3:8.1f} us/step  {n_envs/ms/1e3:8.2f} M env-steps/s  contacts={counts.float().mean().item():.1f}", flush=True)
    sim.close()

if __name__ == "__main__":
    robot = sys.argv[1] if len(sys.argv) > 1 else "nao"
    for n_envs in (1024, 3072, 4096, 12288, 16384):
        run(robot, n_envs)
    run(robot, 4096, solver_iters=0)
    run(robot, 4096, solver_iters=100)
    run(robot, 4096, enable_ground=False)
    run(robot, 4096, enable_ground=False, solver_iters=0)
    run(robot, 12288, solver_iters=0)
    run(robot, 12288, enable_ground=False)
    run(robot, 12288, enable_ground=False, solver_iters=0)
