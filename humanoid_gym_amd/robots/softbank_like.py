"""Loader for the NAO / Pepper / Romeo stand-in models (see robots/__init__.py).

Posture tables: the reference reads ``NaoPosture('Stand')`` / ``PepperPosture('Stand')``
from qibullet (nao_env.py:24-27, pepper_env.py:22-25); qibullet is absent here,
so the values below are the NAOqi "Stand" posture as publicly documented,
rounded -- approximate, like the models themselves.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from ..model import FlatModel
from ..urdf import load_urdf, quat_to_mat

_HERE = os.path.dirname(os.path.abspath(__file__))

NAO_STAND = {
    "HeadYaw": 0.0, "HeadPitch": -0.17,
    "LShoulderPitch": 1.44, "LShoulderRoll": 0.22, "LElbowYaw": -1.2, "LElbowRoll": -0.42, "LWristYaw": 0.1, "LHand": 0.29,
    "RShoulderPitch": 1.44, "RShoulderRoll": -0.22, "RElbowYaw": 1.2, "RElbowRoll": 0.42, "RWristYaw": 0.1, "RHand": 0.29,
    "LHipYawPitch": -0.17, "LHipRoll": 0.1, "LHipPitch": 0.13, "LKneePitch": -0.09, "LAnklePitch": 0.09, "LAnkleRoll": -0.13,
    "RHipYawPitch": -0.17, "RHipRoll": -0.1, "RHipPitch": 0.13, "RKneePitch": -0.09, "RAnklePitch": 0.09, "RAnkleRoll": 0.13,
}
PEPPER_STAND = {
    "KneePitch": 0.0, "HipPitch": -0.04, "HipRoll": 0.0, "HeadYaw": 0.0, "HeadPitch": -0.2,
    "LShoulderPitch": 1.58, "LShoulderRoll": 0.14, "LElbowYaw": -1.23, "LElbowRoll": -0.52, "LWristYaw": 0.0, "LHand": 0.59,
    "RShoulderPitch": 1.58, "RShoulderRoll": -0.14, "RElbowYaw": 1.23, "RElbowRoll": 0.52, "RWristYaw": 0.0, "RHand": 0.59,
}
ROMEO_STAND = {
    "LShoulderPitch": 1.4, "RShoulderPitch": 1.4, "LShoulderYaw": 0.2, "RShoulderYaw": -0.2,
    "LElbowYaw": -0.4, "RElbowYaw": 0.4,
    "LHipPitch": -0.2, "RHipPitch": -0.2, "LKneePitch": 0.4, "RKneePitch": 0.4, "LAnklePitch": -0.2, "RAnklePitch": -0.2,
}


def _lowest_point(m: FlatModel) -> float:
    """Lowest collision-sphere bottom (world z) at the reset pose."""
    n = m.n_links
    Rb, xb = quat_to_mat(m.base_quat0), m.base_pos0
    Rw, xw = [None] * n, [None] * n
    for i in range(n):
        p = m.parent[i]
        Rp, xp = (Rb, xb) if p < 0 else (Rw[p], xw[p])
        a, q = m.axis[i], -m.q_init[i]
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        Rq = np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)
        E = Rq @ m.R0[i]
        Rw[i] = Rp @ E.T
        xw[i] = xp + Rp @ m.d[i] + Rw[i] @ m.e[i]
    lo = np.inf
    for s in range(len(m.sphere_link)):
        b = m.sphere_link[s]
        R, x = (Rb, xb) if b < 0 else (Rw[b], xw[b])
        lo = min(lo, (x + R @ m.sphere_pos[s])[2] - m.sphere_radius[s])
    return float(lo)


def load_softbank_like(name: str, urdf_path: Optional[str] = None, hands: bool = False) -> FlatModel:
    """hands=True (nao, romeo): the stand-in that also has the finger / thumb joints of SoftBank's hands --
    they are simulated dofs of the reference's robots (romeo_env.py:24-26 filters them out of the CONTROLLED
    set only), 16 for NAO (42 dofs in all), 24 for Romeo (61)."""
    from . import NAO_JOINTS, PEPPER_JOINTS
    posture = {"nao": NAO_STAND, "pepper": PEPPER_STAND, "romeo": ROMEO_STAND}[name]
    if hands and name not in ("nao", "romeo"):
        raise KeyError(f"no hands variant of {name!r}")
    path = urdf_path or os.path.join(_HERE, f"{name}_hands_like.urdf" if hands else f"{name}_like.urdf")
    probe = load_urdf(path, name=name)
    q_init = {k: v for k, v in posture.items() if k in probe.joint2Index}
    # <mimic> joints start where the joint they follow puts them (qibullet drives them together)
    for child, (parent, mult, off) in (probe.meta.get("mimic") or {}).items():
        if child in probe.joint2Index and child not in q_init:
            k = probe.dof_of(child)
            q_init[child] = float(np.clip(mult * q_init.get(parent, 0.0) + off, probe.lower[k], probe.upper[k]))
    m = load_urdf(path, base_position=(0, 0, 0), q_init=q_init, self_collision=True,
                  default_max_force=100000.0, name=name)
    # spawn like qibullet does: upright at the origin, soles 1 mm above the ground plane
    m.base_pos0 = m.base_pos0 + np.array([0.0, 0.0, 0.001 - _lowest_point(m)])
    if name == "nao":
        m.set_act_joints(NAO_JOINTS)                       # nao_env.py:30-31
    elif name == "pepper":
        m.set_act_joints(PEPPER_JOINTS)                    # pepper_env.py:28-39
    else:                                                  # romeo_env.py:24-26
        m.set_act_joints([n for n in m.dof_names if "Finger" not in n and "Thumb" not in n])
    m.check()
    return m
