"""Replays a sample dumped by a fuzzer (FUZZ_DUMP=<dir>) on the kernel and the oracle under variations of the config, to
find which part of the step a difference comes from.  usage (GPU box): python tools/fuzz_replay.py <dir>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from fuzz_common import OracleSim, HgSim
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
if "mass" in z.files:          # fuzz_trees.py scales the link masses of some shapes
    model.mass, model.inertia = z["mass"], z["inertia"]
base_cmd = z["base_cmd"] if "base_cmd" in z.files and len(z["base_cmd"]) else None
n = model.n_links
print(meta["tag"], "env", e)
qb = before[e, 13:13 + n]
viol = [i for i in range(n) if model.limited[i] and (qb[i] <= model.lower[i] or qb[i] >= model.upper[i])]
print("joints at or beyond a bound before the step:", viol, "parents", [int(model.parent[i]) for i in viol])


def run(label, **over):
    cfg = SimConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["cfg"].items()})
    n_sub = over.pop("n_substeps", sub)
    for k, v in over.items():
        setattr(cfg, k, v)
    g, c = HgSim(model, cfg, before.shape[0], "cuda:0"), OracleSim(model, cfg, before.shape[0])
    c.state[:] = before
    if base_cmd is not None:
        g.set_base_velocity(torch.tensor(base_cmd, dtype=torch.float32))
    g.set_state(torch.tensor(before, dtype=torch.float32))
    o = c.step(act, joint_sel=sel, n_substeps=n_sub, base_cmd=base_cmd)
    out = g.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"), joint_sel=sel, n_substeps=n_sub)
    q = out[0].cpu().numpy()
    st = g.get_state().cpu().numpy()
    tau = g.joint_torques().cpu().numpy() if hasattr(g, "joint_torques") else None
    dq = np.abs(q[e] - o["q"][e])
    dqd = np.abs(st[e, 13 + n:13 + 2 * n] - c.qd[e])
    bad = np.nonzero(dqd > 1e-2)[0]
    dtau = np.abs(tau[e] - o["motor_torque"][e])
    jt = int(dtau.argmax())
    print(f"{label:44s} max |dq| {dq.max():.3g} (column {dq.argmax()}), max |dqd| {dqd.max():.3g}; links with |dqd| > 1e-2: {bad.tolist()}; "
          f"contacts {int(o['n_contacts'][e])}; torque: link {jt} kernel {tau[e][jt]:.6g} oracle {o['motor_torque'][e][jt]:.6g} "
          f"(bound {model.max_force[jt]:.3g}, q {st[e, 13 + jt]:.5f} in [{model.lower[jt]:.5f}, {model.upper[jt]:.5f}])")
    g.close()
    return o, st


run("as dumped")
run("limits off", enable_limits=False)
run("self contacts off", enable_self=False)
run("ground off", enable_ground=False)
run("no contacts at all", enable_ground=False, enable_self=False)
run("friction 0", friction=0.0)
for it in (1, 2, 3, 4, 8, 50, 150):
    run(f"{it} sweeps", solver_iters=it)
for mc in (4, 6, 8, 12):
    run(f"max_contacts {mc}", max_contacts=mc)
if sub > 1:
    for s2 in range(1, sub):
        run(f"{s2} substeps", n_substeps=s2)
