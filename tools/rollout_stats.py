"""Closed-loop statistics of free-running rollouts, kernel against oracle: N envs x T steps per robot, the same random
targets on both sides (new targets every 20 steps, per env), NO re-synchronisation.  Single trajectories separate
(contacts make the dynamics chaotic; one-step parity is what the tests and fuzzers hold), so what is compared is what an RL
user sees: per-step means over the envs of base height, joint-rate rms, collision rate and contact count, and the
distribution of the final base height (two-sample Kolmogorov-Smirnov).  A systematic bias of the FP32 kernel would show here.
usage (GPU box): python tools/rollout_stats.py [n_envs] [n_steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(1, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from scipy.stats import ks_2samp
from _oracle import OracleSim
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd._capi import HgSim


def rollout(robot, n_envs, n_steps, seed=0):
    model, cfg = load_robot(robot)
    sub = 4 if robot == "dancer" else 1
    rng = np.random.default_rng(seed)
    gpu, cpu = HgSim(model, cfg, n_envs, "cuda:0"), OracleSim(model, cfg, n_envs)
    idx = np.asarray(model.act_joint)
    lo, hi = np.maximum(model.lower[idx], -1.0), np.minimum(model.upper[idx], 1.0)
    n = model.n_links
    series = {k: ([], []) for k in ("height", "qd_rms", "collision", "contacts")}
    for t in range(n_steps):
        if t % 20 == 0:
            act = rng.uniform(lo, hi, (n_envs, len(idx))).astype(np.float32)
        o = cpu.step(act.astype(np.float64), n_substeps=sub)
        _, _, _, col = gpu.step(torch.tensor(act, device="cuda:0"), n_substeps=sub)
        if t % 10 == 9:
            st = gpu.get_state().cpu().numpy()
            _, counts, _ = gpu.contact_pairs()
            series["height"][0].append(st[:, 2].mean()); series["height"][1].append(cpu.state[:, 2].mean())
            series["qd_rms"][0].append(np.sqrt((st[:, 13 + n:13 + 2 * n] ** 2).mean())); series["qd_rms"][1].append(np.sqrt((cpu.qd ** 2).mean()))
            series["collision"][0].append(col.cpu().numpy().mean()); series["collision"][1].append(o["collision"].mean())
            series["contacts"][0].append(counts.cpu().numpy().mean()); series["contacts"][1].append(o["n_contacts"].mean())
    st = gpu.get_state().cpu().numpy()
    ks = ks_2samp(st[:, 2], cpu.state[:, 2])
    finite = bool(np.isfinite(st).all() and np.isfinite(cpu.state).all())
    gpu.close()
    out = dict(robot=robot, n_envs=n_envs, n_steps=n_steps, ks_stat=float(ks.statistic), ks_p=float(ks.pvalue), finite=finite)
    for k, (g, c) in series.items():
        g, c = np.asarray(g), np.asarray(c)
        out[k] = dict(kernel_last=float(g[-1]), oracle_last=float(c[-1]), worst_abs=float(np.abs(g - c).max()),
                      worst_rel=float((np.abs(g - c) / np.maximum(np.abs(c), 1e-3)).max()))
    return out


def run(n_envs=1024, n_steps=480, robots=("nao_hands", "romeo_hands", "nao", "pepper", "romeo", "dancer")):
    res = []
    for robot in robots:
        r = rollout(robot, n_envs, n_steps)
        res.append(r)
        print(f"{robot:11s} {n_envs} envs x {n_steps} steps: final-height KS {r['ks_stat']:.3f} (p {r['ks_p']:.2f}); "
              + "; ".join(f"{k} {r[k]['kernel_last']:.4g} / {r[k]['oracle_last']:.4g} (worst gap {r[k]['worst_abs']:.2g})" for k in ("height", "qd_rms", "collision", "contacts")),
              flush=True)
    return res


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 480,
        *([tuple(sys.argv[3:])] if len(sys.argv) > 3 else []))
