"""How much of the finger-joint models' step time goes to joint-limit rows: the bench's NAO-with-fingers replay and the
Romeo-with-hands random-target workload with enable_limits on / off, and how many joints sit at or beyond a bound.
usage (GPU box): python tools/hands_limits_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd._capi import HgSim
from humanoid_gym_amd.workloads import nao_replay_joints

dev = torch.device("cuda", 0)
for robot, n_envs in (("nao_hands", 4096), ("romeo_hands", 8192), ("nao", 4096)):
    for limits in (True, False):
        model, cfg = load_robot(robot)
        cfg.enable_limits = limits
        if robot == "nao_hands":
            rows, sel = bench.synthetic_replay(model)
            rows, sel = bench.with_mimic_columns(model, rows, sel, nao_replay_joints())
        elif robot == "nao":
            rows, sel = bench.synthetic_replay(model)
        else:
            rng = np.random.default_rng(3)
            idx = np.asarray(model.act_joint)
            rows, sel = rng.uniform(model.lower[idx], model.upper[idx], (240, len(idx))).astype(np.float32), None
        sim = HgSim(model, cfg, n_envs, dev)
        rows_d = torch.tensor(rows, device=dev)
        phase = torch.arange(n_envs, device=dev) % 240
        act = torch.stack([rows_d[(t + phase) % 240] for t in range(240)]).contiguous()
        for t in range(40):
            sim.step(act[t % 240], joint_sel=sel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(40, 240):
            sim.step(act[t % 240], joint_sel=sel)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
        st = sim.get_state().cpu().numpy()
        n = model.n_links
        q = st[:, 13:13 + n]
        lim = np.asarray(model.limited, bool)
        at = ((q <= model.lower) | (q >= model.upper)) & lim
        print(f"{robot:12s} {n_envs} envs, limits {'on ' if limits else 'off'}: {ms:.4f} ms/step = {n_envs / ms / 1e3:.2f} M env-steps/s; "
              f"envs with a joint at or beyond a bound: {at.any(1).mean():.2%}, such joints per env: {at.sum(1).mean():.2f}", flush=True)
        sim.close()
