"""Generates the committed fixtures under tests/golden/.  Run in the build container:

    python tests/golden/make_golden.py

1. dancer_layout.json -- the INTEGER half of the contract, derived from the
   reference's own URDF with a bare xml.etree walk (deliberately independent of
   humanoid_gym_amd.urdf): joint / link names in depth-first declaration order
   (= PyBullet's joint indices, dancer_env.py:73-79), parent indices, axes, and
   the reference env's name lists for nao/pepper (nao_env.py:30-31,
   pepper_env.py:28-39, read from the reference's source text by regex).
2. oracle_*.npz -- short trajectories produced by THIS repo's FP64 oracle
   (oracle/hg_oracle.c).  They pin the oracle against accidental change and let
   the GPU box check the HIP path without /root/reference; they are NOT
   PyBullet outputs (parity with PyBullet is unpinned).
"""
import json
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def dancer_layout():
    root = ET.parse(os.path.join(REF, "humanoid_gym/envs/assets/dancer/dancer_urdf_model.URDF")).getroot()
    joints = [(j.get("name"), j.get("type"), j.find("parent").get("link"), j.find("child").get("link"),
               [float(x) for x in j.find("axis").get("xyz").split()],
               [float(x) for x in j.find("origin").get("rpy").split()]) for j in root.findall("joint")]
    children = {j[3] for j in joints}
    links = [l.get("name") for l in root.findall("link")]
    base = [l for l in links if l not in children]
    assert len(base) == 1
    order = []

    def visit(link):
        for j in joints:
            if j[2] == link:
                order.append(j)
                visit(j[3])
    visit(base[0])
    link_index = {j[3]: i for i, j in enumerate(order)}
    masses = {l.get("name"): float(l.find("inertial").find("mass").get("value")) for l in root.findall("link")}
    return {
        "base_link": base[0],
        "joint_names": [j[0] for j in order],
        "link_names": [j[3] for j in order],
        "joint_types": [j[1] for j in order],
        "parent_index": [link_index.get(j[2], -1) for j in order],
        "axis": [j[4] for j in order],
        "origin_rpy": [j[5] for j in order],
        "total_mass": sum(masses.values()),
        # computed in the survey from URDF + meshes at the reset pose
        "reset_com": [0.3185, -0.4976, 0.2988], "reset_lowest_z": 0.0352,
    }


def env_name_lists():
    out = {}
    src = open(os.path.join(REF, "humanoid_gym/envs/nao_env.py")).read()
    m = re.search(r"self\.joint_names = \[(.*?)\]", src, re.S)
    out["nao"] = re.findall(r"'(\w+)'", m.group(1))
    src = open(os.path.join(REF, "humanoid_gym/envs/pepper_env.py")).read()
    m = re.search(r"self\.joint_names = \[(.*?)\]", src, re.S)
    out["pepper"] = re.findall(r"'(\w+)'", m.group(1))
    src = open(os.path.join(REF, "humanoid_gym/__init__.py")).read()
    out["env_ids"] = re.findall(r"id='([\w-]+)'", src)
    return out


def oracle_trajectories():
    from _oracle import OracleSim
    from humanoid_gym_amd.robots import load_robot
    from humanoid_gym_amd.workloads import nao_replay_joints, pack_rows, synthetic_h5_trajectory
    # dancer: random targets U(-1,1), default_rng(4), held 12 steps each, ground + self contact on (dancer_env.py:60).
    # ONE-STEP vectors: the state before every step is rounded to float32 (what a GPU buffer can hold exactly) and
    # stored, so that the test steps the kernel once from each stored state -- a trajectory of a robot landing on its
    # hull vertices is chaotic, one step is not.
    model, cfg = load_robot("dancer")
    sim = OracleSim(model, cfg, 2)
    rng = np.random.default_rng(4)
    acts, qs, ncs, masks, states, pairs = [], [], [], [], [], []
    for t in range(96):
        if t % 12 == 0:
            a = rng.uniform(-1, 1, (2, model.n_act)).astype(np.float32).astype(np.float64)
        sim.state[:] = sim.state.astype(np.float32)
        states.append(sim.state.copy())
        o = sim.step(a)
        acts.append(a.copy()); qs.append(o["q"].copy()); ncs.append(o["n_contacts"].copy()); masks.append(o["link_mask"].copy())
        pairs.append(o["pairs"].copy())
    np.savez_compressed(os.path.join(HERE, "oracle_dancer_random.npz"), actions=np.array(acts, np.float32),
                        state=np.array(states, np.float32), q=np.array(qs), n_contacts=np.array(ncs),
                        link_mask=np.array(masks), pairs=np.array(pairs, np.int32), final_state=sim.state)
    # nao: the replay workload, contact-free (the 1e-4 rad case) and with ground contact
    model, cfg = load_robot("nao")
    l, r = synthetic_h5_trajectory(model, T=240, seed=1234)
    rows = pack_rows(l, r)
    sel = np.array([model.dof_of(n) for n in nao_replay_joints()], np.int32)
    for tag, kw in (("free", dict(enable_ground=False, enable_self=False, gravity=(0.0, 0.0, 0.0))), ("ground", {})):
        model, cfg = load_robot("nao")
        for k, v in kw.items():
            setattr(cfg, k, v)
        sim = OracleSim(model, cfg, 2)
        qs, ncs = [], []
        for t in range(240):
            o = sim.step(rows[(t + np.array([0, 37])) % 240], joint_sel=sel)
            qs.append(o["q"].copy()); ncs.append(o["n_contacts"].copy())
        np.savez_compressed(os.path.join(HERE, f"oracle_nao_replay_{tag}.npz"), rows=rows, sel=sel,
                            q=np.array(qs), n_contacts=np.array(ncs), final_state=sim.state)


if __name__ == "__main__":
    lay = dancer_layout()
    lay.update(env_name_lists())
    with open(os.path.join(HERE, "dancer_layout.json"), "w") as f:
        json.dump(lay, f, indent=1)
    oracle_trajectories()
    print("golden fixtures written to", HERE)
