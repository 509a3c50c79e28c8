"""At which sweep would Bullet's residual early-out leave the solver?  (measure before building.)

Runs the FP64 oracle on the workloads of the bench (NAO h5-layout replay, 26 and 42 dofs) and on Dancer with random
targets and 4 substeps (dancer_env.py:63-65: 150 sweeps), with the config unchanged (every sweep runs), and records per
physics step the first sweep after which the largest squared velocity residual of the sweep was <= the probe threshold
(PyBullet's solverResidualThreshold default 1e-7 [UPSTREAM-RECALL]).  CPU only.
usage: python tools/residual_histogram.py [threshold]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from _oracle import OracleSim, lib
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd.workloads import nao_replay_joints

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-7
L = lib()
L.hgo_residual_probe.argtypes = [C.c_double]; L.hgo_residual_probe.restype = None
L.hgo_residual_histogram.argtypes = [C.POINTER(C.c_long), C.c_int]; L.hgo_residual_histogram.restype = None


def report(name, iters):
    h = (C.c_long * 1026)()
    L.hgo_residual_histogram(h, 1026)
    h = np.array(h[:iters + 2], dtype=np.int64)
    tot = h.sum()
    cum = np.cumsum(h)
    med = int(np.searchsorted(cum, 0.5 * tot)); p90 = int(np.searchsorted(cum, 0.9 * tot)); p99 = int(np.searchsorted(cum, 0.99 * tot))
    never = h[iters + 1] / tot
    mean = float((np.arange(len(h)) * h).sum() / tot)
    print(f"{name}: {tot} physics steps, {iters} sweeps configured; first sweep with residual^2 <= {thr:g}: "
          f"median {med}, mean {mean:.1f}, 90 % {p90}, 99 % {p99}; never within the configured sweeps: {never:.2%}")
    print("   histogram (sweep: steps): " + ", ".join(f"{k}: {v}" for k, v in enumerate(h) if v))


for robot in ("nao", "nao_hands"):
    model, cfg = load_robot(robot)
    rows, sel = bench.synthetic_replay(model)
    if robot == "nao_hands":
        rows, sel = bench.with_mimic_columns(model, rows, sel, nao_replay_joints())
    n_envs = 64
    sim = OracleSim(model, cfg, n_envs)
    phase = np.arange(n_envs) % bench.T_TRAJ
    for t in range(60):                       # the fall onto the ground is not part of the statistic
        sim.step(rows[(t + phase) % bench.T_TRAJ], joint_sel=sel)
    L.hgo_residual_probe(thr)
    for t in range(60, 300):
        sim.step(rows[(t + phase) % bench.T_TRAJ], joint_sel=sel)
    report(f"{robot} replay (bench workload)", cfg.solver_iters)

for robot, sub in (("dancer", 4), ("pepper", 1), ("romeo", 1)):
    model, cfg = load_robot(robot)
    n_envs = 64
    sim = OracleSim(model, cfg, n_envs)
    rng = np.random.default_rng(4)
    idx = np.asarray(model.act_joint)
    lo, hi = (-1.0, 1.0) if robot == "dancer" else (model.lower[idx], model.upper[idx])
    L.hgo_residual_probe(thr)
    for t in range(120):
        sim.step(rng.uniform(lo, hi, (n_envs, len(idx))), n_substeps=sub)
    report(f"{robot} random targets x {sub} substeps", cfg.solver_iters)
L.hgo_residual_probe(0.0)
