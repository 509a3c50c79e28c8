"""One-step kernel-vs-oracle error of a robot grouped by contact count (diagnostics for the overflow-row solver).
usage (GPU box): python tools/ovf_debug.py [robot] [key=value ...]"""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from _oracle import OracleSim
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd._capi import HgSim

robot = sys.argv[1] if len(sys.argv) > 1 else "nao_hands"
model, cfg = load_robot(robot)
for a in sys.argv[2:]:
    k, v = a.split("=", 1)
    setattr(cfg, k, type(getattr(cfg, k))(ast.literal_eval(v)))
N = 64
gpu, cpu = HgSim(model, cfg, N, "cuda:0"), OracleSim(model, cfg, N)
rng = np.random.default_rng(5)
idx = np.asarray(model.act_joint)
lo, hi = np.maximum(model.lower[idx], -1.2), np.minimum(model.upper[idx], 1.2)
err_by_nc, self_by_nc = {}, {}
for t in range(150):
    if t % 10 == 0:
        act = rng.uniform(lo, hi, (N, len(idx))).astype(np.float32).astype(np.float64)
    cpu.state[:] = cpu.state.astype(np.float32)
    gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
    o = cpu.step(act)
    gq = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"))[0].cpu().numpy()
    for e in range(N):
        nc = int(o["n_contacts"][e])
        err_by_nc.setdefault(nc, []).append(float(np.abs(gq[e] - o["q"][e]).max()))
        if nc:
            last_self = o["pairs"][e][nc - 1][1] >= -1
            self_by_nc.setdefault((nc, bool(last_self)), []).append(err_by_nc[nc][-1])
for nc in sorted(err_by_nc):
    v = np.array(err_by_nc[nc])
    print(f"nc={nc}: {len(v)} samples, max err {v.max():.3g}, median {np.median(v):.3g}, > 1e-4: {(v > 1e-4).sum()}")
for k in sorted(self_by_nc):
    v = np.array(self_by_nc[k])
    print(f"nc={k[0]} last contact link-link={k[1]}: {len(v)} samples, max {v.max():.3g}, > 1e-4: {(v > 1e-4).sum()}")
