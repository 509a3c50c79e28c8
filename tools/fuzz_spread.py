"""For a sample dumped by a fuzzer: how far does the ORACLE's own answer move under float32-sized noise inside its step
(OracleSim.noise_spread, the probe behind the explained-outlier rule), at several noise sizes, and with one sweep more or
fewer?  CPU only.  usage: python tools/fuzz_spread.py <dump dir>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _oracle import OracleSim
from humanoid_gym_amd.model import SimConfig
from humanoid_gym_amd.urdf import load_urdf
from humanoid_gym_amd.robots import load_robot

d = sys.argv[1]
meta = json.load(open(os.path.join(d, "sample.json")))
z = np.load(os.path.join(d, "sample.npz"))
before, act, sub, e = z["before"], z["act"], int(z["sub"]), int(z["env"])
sel = z["sel"] if meta["has_sel"] else None
if meta["urdf"]:
    model = load_urdf(os.path.join(d, "sample.urdf"), self_collision=meta["cfg"]["enable_self"], name="tree")
else:
    model, _ = load_robot(meta["name"])
    model.lower, model.upper = model.lower.astype(np.float32).astype(np.float64), model.upper.astype(np.float32).astype(np.float64)
if "max_force" in z.files:
    model.max_force, model.max_vel = z["max_force"], z["max_vel"]
if "mass" in z.files:
    model.mass, model.inertia = z["mass"], z["inertia"]
base_cmd = z["base_cmd"] if "base_cmd" in z.files and len(z["base_cmd"]) else None
cfg = SimConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["cfg"].items()})
n = model.n_links
print(meta["tag"], "env", e, "| kernel - oracle as dumped:", float(np.abs(z["kernel_q"][e] - z["oracle_q"][e]).max()), "rad")
c = OracleSim(model, cfg, before.shape[0])
for eps in (1e-8, 1e-7, 3e-7, 1e-6, 3e-6):
    sp = c.noise_spread(before[e], act[e], joint_sel=sel, n_substeps=sub, base_cmd_row=None if base_cmd is None else base_cmd[e], trials=256, eps=eps)
    print(f"noise {eps:.0e}: oracle spread q {float(np.max(sp['q'])):.3g} rad, qd {float(np.max(sp['qd'])):.3g} rad/s, base {float(np.max(sp['base'])):.3g}, torque {float(np.max(sp['tau_rel'])):.3g}")
ref = None
for it in (cfg.solver_iters - 1, cfg.solver_iters, cfg.solver_iters + 1, 2 * cfg.solver_iters, 50, 150):
    if it < 0:
        continue
    import dataclasses
    c2 = OracleSim(model, dataclasses.replace(cfg, solver_iters=it), before.shape[0])
    c2.state[:] = before
    o = c2.step(act, joint_sel=sel, n_substeps=sub, base_cmd=base_cmd)
    if it == cfg.solver_iters:
        ref = o["q"][e].copy()
    print(f"{it:4d} sweeps: oracle q[worst column] {o['q'][e][int(np.abs(z['kernel_q'][e] - z['oracle_q'][e]).argmax())]:.6f}, max |qd| {float(np.abs(c2.qd[e]).max()):.3g}, contacts {o['pairs'][e][:int(o['n_contacts'][e])].tolist()}")
