"""Launch-bound regime: steps of a few envs, eager launches vs one HIP graph of K captured steps.
usage: graph_replay.py [robot]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from humanoid_gym_amd._capi import HgSim
from humanoid_gym_amd.robots import load_robot

robot = sys.argv[1] if len(sys.argv) > 1 else "nao"
model, cfg = load_robot(robot)
K = 16
for n_envs in (1, 64, 1024, 4096):
    sim = HgSim(model, cfg, n_envs, "cuda:0")
    idx = np.asarray(model.act_joint)
    acts = torch.tensor(np.random.default_rng(0).uniform(model.lower[idx], model.upper[idx], (K, n_envs, model.n_act)),
                        dtype=torch.float32, device="cuda:0")
    for k in range(K):
        sim.step(acts[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(20):
        for k in range(K):
            sim.step(acts[k])
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / (20 * K)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(K):
            sim.step(acts[k])
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(20):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / (20 * K)
    print(f"{robot} {n_envs:5d} envs: eager {eager * 1e6:7.1f} us/step, graph of {K} steps {graph * 1e6:7.1f} us/step", flush=True)
    sim.close()
