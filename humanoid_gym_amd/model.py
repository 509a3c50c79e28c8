"""Flat robot model: the host-side mirror of ``include/hg_model.h``.

The reference never holds a robot description of its own: it asks PyBullet to
build one (``p.loadURDF`` in /root/reference/humanoid_gym/envs/dancer_env.py:58-60,
``spawnNao`` & co. in nao_env.py:20, pepper_env.py:18, romeo_env.py:17) and then
queries joint names, indices and limits back (dancer_env.py:73-82,
nao_env.py:35-43).  ``FlatModel`` is the structure-of-arrays result of that
load: topology, per-link constants in centre-of-mass/principal-axis frames,
collision spheres, plus the name <-> index maps the env classes expose.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

HG_MAX_LINKS = 62
HG_MAX_SPHERES = 1024
HG_MAX_PAIRS = 1024
HG_MAX_CONTACTS = 12
HG_MAX_ROWS = 128
HG_STATE_BASE = 13

SPHERE_GROUND = 1
SPHERE_SELF = 2


class HgModel(C.Structure):
    """ctypes mirror of ``struct hg_model`` (include/hg_model.h)."""

    _fields_ = [
        ("n_links", C.c_int32),
        ("n_spheres", C.c_int32),
        ("n_pairs", C.c_int32),
        ("n_act", C.c_int32),
        ("parent", C.c_int32 * HG_MAX_LINKS),
        ("act_joint", C.c_int32 * HG_MAX_LINKS),
        ("R0", (C.c_double * 9) * HG_MAX_LINKS),
        ("axis", (C.c_double * 3) * HG_MAX_LINKS),
        ("d", (C.c_double * 3) * HG_MAX_LINKS),
        ("e", (C.c_double * 3) * HG_MAX_LINKS),
        ("mass", C.c_double * HG_MAX_LINKS),
        ("inertia", (C.c_double * 3) * HG_MAX_LINKS),
        ("joint_damping", C.c_double * HG_MAX_LINKS),
        ("lower", C.c_double * HG_MAX_LINKS),
        ("upper", C.c_double * HG_MAX_LINKS),
        ("max_force", C.c_double * HG_MAX_LINKS),
        ("max_vel", C.c_double * HG_MAX_LINKS),
        ("q_init", C.c_double * HG_MAX_LINKS),
        ("joint_limited", C.c_int32 * HG_MAX_LINKS),
        ("base_mass", C.c_double),
        ("base_inertia", C.c_double * 3),
        ("base_pos0", C.c_double * 3),
        ("base_quat0", C.c_double * 4),
        ("frame_pos", (C.c_double * 3) * (HG_MAX_LINKS + 1)),
        ("frame_quat", (C.c_double * 4) * (HG_MAX_LINKS + 1)),
        ("sphere_link", C.c_int32 * HG_MAX_SPHERES),
        ("sphere_flags", C.c_int32 * HG_MAX_SPHERES),
        ("sphere_pos", (C.c_double * 3) * HG_MAX_SPHERES),
        ("sphere_radius", C.c_double * HG_MAX_SPHERES),
        ("sphere_half", (C.c_double * 3) * HG_MAX_SPHERES),
        ("pair_a", C.c_int32 * HG_MAX_PAIRS),
        ("pair_b", C.c_int32 * HG_MAX_PAIRS),
    ]


class HgConfig(C.Structure):
    """ctypes mirror of ``struct hg_config`` (include/hg_model.h)."""

    _fields_ = [
        ("dt", C.c_double),
        ("gravity", C.c_double * 3),
        ("kp", C.c_double),
        ("kd", C.c_double),
        ("contact_erp", C.c_double),
        ("friction", C.c_double),
        ("contact_margin", C.c_double),
        ("lin_damping", C.c_double),
        ("ang_damping", C.c_double),
        ("max_coord_velocity", C.c_double),
        ("solver_iters", C.c_int32),
        ("max_contacts", C.c_int32),
        ("enable_ground", C.c_int32),
        ("enable_self", C.c_int32),
        ("clamp_targets", C.c_int32),
        ("enable_limits", C.c_int32),
        ("limit_max_impulse", C.c_double),
        ("residual_threshold", C.c_double),
        ("store_torques", C.c_int32),
        ("store_contacts", C.c_int32),
    ]


@dataclass
class SimConfig:
    """World parameters.  Dancer values are the literals of
    dancer_env.py:63-65; the qibullet robots use PyBullet's defaults
    (g = -9.81, 50 solver iterations, 1/240 s)."""

    dt: float = 1.0 / 240.0
    gravity: tuple = (0.0, 0.0, -9.81)
    kp: float = 0.1
    kd: float = 1.0
    contact_erp: float = 0.2
    friction: float = 0.5
    contact_margin: float = 0.02
    lin_damping: float = 0.04
    ang_damping: float = 0.04
    max_coord_velocity: float = 100.0    # btMultiBody m_maxCoordinateVelocity [UPSTREAM-RECALL]
    solver_iters: int = 50
    max_contacts: int = 8
    enable_ground: bool = True
    enable_self: bool = False
    clamp_targets: bool = True
    enable_limits: bool = True           # btMultiBodyJointLimitConstraint rows for revolute joints
    limit_max_impulse: float = 100.0     # btMultiBodyConstraint m_maxAppliedImpulse [UPSTREAM-RECALL]
    residual_threshold: float = 0.0      # > 0: leave the sweep loop once a sweep's largest squared velocity residual is <= this
                                         # (PyBullet's solverResidualThreshold default is 1e-7 [UPSTREAM-RECALL]); 0: every sweep
    store_torques: bool = True           # keep the applied motor torques of each physics step for joint_torques()
    store_contacts: bool = True          # keep the contact link pairs / counts / link masks of each step for contact_pairs()

    def to_c(self) -> HgConfig:
        c = HgConfig()
        c.dt = self.dt
        c.gravity[:] = list(self.gravity)
        c.kp, c.kd = self.kp, self.kd
        c.contact_erp = self.contact_erp
        c.friction = self.friction
        c.contact_margin = self.contact_margin
        c.lin_damping, c.ang_damping = self.lin_damping, self.ang_damping
        c.max_coord_velocity = self.max_coord_velocity
        c.solver_iters = int(self.solver_iters)
        c.max_contacts = int(self.max_contacts)
        c.enable_ground = int(self.enable_ground)
        c.enable_self = int(self.enable_self)
        c.clamp_targets = int(self.clamp_targets)
        c.enable_limits = int(self.enable_limits)
        c.limit_max_impulse = self.limit_max_impulse
        c.residual_threshold = float(self.residual_threshold)
        c.store_torques = int(self.store_torques)
        c.store_contacts = int(self.store_contacts)
        return c


@dataclass
class FlatModel:
    name: str
    base_link: str
    # PyBullet-style maps over ALL joints of the URDF (fixed ones included),
    # numbered depth-first in declaration order (dancer_env.py:73-79).
    joint_names_all: List[str]
    link_names_all: List[str]          # child link of joint i
    joint_types_all: List[str]
    joint_dof: List[int]               # joint index -> dof index, -1 for fixed
    # dynamic part (one entry per dof)
    dof_joint: List[int]               # dof index -> joint index
    parent: np.ndarray
    R0: np.ndarray
    axis: np.ndarray
    d: np.ndarray
    e: np.ndarray
    mass: np.ndarray
    inertia: np.ndarray
    joint_damping: np.ndarray
    lower: np.ndarray
    upper: np.ndarray
    max_force: np.ndarray
    max_vel: np.ndarray
    q_init: np.ndarray
    base_mass: float
    base_inertia: np.ndarray
    base_pos0: np.ndarray
    base_quat0: np.ndarray
    frame_pos: np.ndarray              # [n+1, 3]
    frame_quat: np.ndarray             # [n+1, 4]
    sphere_link: np.ndarray
    sphere_flags: np.ndarray
    sphere_pos: np.ndarray
    sphere_radius: np.ndarray
    pair_a: np.ndarray
    pair_b: np.ndarray
    act_joint: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    meta: Dict = field(default_factory=dict)
    sphere_half: Optional[np.ndarray] = None      # [n_spheres, 3] half axis of a sphere-swept segment, 0 = sphere (None: all spheres)

    def __post_init__(self):
        if self.sphere_half is None or len(self.sphere_half) != len(self.sphere_link):
            self.sphere_half = np.zeros((len(self.sphere_link), 3))

    # ------------------------------------------------------------------
    @property
    def n_links(self) -> int:
        return int(self.parent.shape[0])

    @property
    def limited(self) -> np.ndarray:
        """Per dof: URDF `revolute` with lower < upper (Bullet adds a joint-limit constraint); `continuous` not."""
        t = [self.joint_types_all[j] for j in self.dof_joint]
        return np.array([1 if (t[k] == "revolute" and self.lower[k] < self.upper[k]) else 0
                         for k in range(self.n_links)], np.int32)

    @property
    def n_act(self) -> int:
        return int(self.act_joint.shape[0])

    @property
    def dof_names(self) -> List[str]:
        return [self.joint_names_all[j] for j in self.dof_joint]

    @property
    def dof_link_names(self) -> List[str]:
        return [self.link_names_all[j] for j in self.dof_joint]

    @property
    def joint2Index(self) -> Dict[str, int]:
        return {n: i for i, n in enumerate(self.joint_names_all)}

    @property
    def link2Index(self) -> Dict[str, int]:
        return {n: i for i, n in enumerate(self.link_names_all)}

    def dof_of(self, joint_name: str) -> int:
        k = self.joint_dof[self.joint2Index[joint_name]]
        if k < 0:
            raise KeyError(f"joint {joint_name!r} is fixed")
        return k

    def set_act_joints(self, names: List[str]) -> None:
        self.act_joint = np.array([self.dof_of(n) for n in names], np.int32)

    def total_mass(self) -> float:
        return float(self.base_mass + self.mass.sum())

    # ------------------------------------------------------------------
    def check(self) -> None:
        n = self.n_links
        if not 0 < n <= HG_MAX_LINKS:
            raise ValueError(f"{n} dofs; supported 1..{HG_MAX_LINKS}")
        if len(self.sphere_link) > HG_MAX_SPHERES:
            raise ValueError(f"{len(self.sphere_link)} collision spheres > {HG_MAX_SPHERES}")
        if len(self.pair_a) > HG_MAX_PAIRS:
            raise ValueError(f"{len(self.pair_a)} self-collision pairs > {HG_MAX_PAIRS}")
        for i, p in enumerate(self.parent):
            if not -1 <= p < i:
                raise ValueError("links must be ordered parents-first")

    def to_c(self) -> HgModel:
        self.check()
        m = HgModel()
        n = self.n_links
        m.n_links = n
        m.n_spheres = len(self.sphere_link)
        m.n_pairs = len(self.pair_a)
        m.n_act = self.n_act
        for i in range(n):
            m.parent[i] = int(self.parent[i])
            m.R0[i][:] = self.R0[i].reshape(9).tolist()
            m.axis[i][:] = self.axis[i].tolist()
            m.d[i][:] = self.d[i].tolist()
            m.e[i][:] = self.e[i].tolist()
            m.mass[i] = float(self.mass[i])
            m.inertia[i][:] = self.inertia[i].tolist()
            m.joint_damping[i] = float(self.joint_damping[i])
            m.lower[i] = float(self.lower[i])
            m.upper[i] = float(self.upper[i])
            m.max_force[i] = float(self.max_force[i])
            m.max_vel[i] = float(self.max_vel[i])
            m.q_init[i] = float(self.q_init[i])
            m.joint_limited[i] = int(self.limited[i])
        for k in range(self.n_act):
            m.act_joint[k] = int(self.act_joint[k])
        m.base_mass = float(self.base_mass)
        m.base_inertia[:] = self.base_inertia.tolist()
        m.base_pos0[:] = self.base_pos0.tolist()
        m.base_quat0[:] = self.base_quat0.tolist()
        for i in range(n + 1):
            m.frame_pos[i][:] = self.frame_pos[i].tolist()
            m.frame_quat[i][:] = self.frame_quat[i].tolist()
        for s in range(m.n_spheres):
            m.sphere_link[s] = int(self.sphere_link[s])
            m.sphere_flags[s] = int(self.sphere_flags[s])
            m.sphere_pos[s][:] = self.sphere_pos[s].tolist()
            m.sphere_radius[s] = float(self.sphere_radius[s])
            m.sphere_half[s][:] = np.asarray(self.sphere_half[s], float).tolist()
        for k in range(m.n_pairs):
            m.pair_a[k] = int(self.pair_a[k])
            m.pair_b[k] = int(self.pair_b[k])
        return m

    # ------------------------------------------------------------------
    _ARRAYS = ("parent", "R0", "axis", "d", "e", "mass", "inertia", "joint_damping",
               "lower", "upper", "max_force", "max_vel", "q_init", "base_inertia",
               "base_pos0", "base_quat0", "frame_pos", "frame_quat", "sphere_link",
               "sphere_flags", "sphere_pos", "sphere_radius", "sphere_half", "pair_a", "pair_b",
               "act_joint")
    _INT = ("parent", "sphere_link", "sphere_flags", "pair_a", "pair_b", "act_joint")

    def to_json(self) -> str:
        out = {
            "name": self.name, "base_link": self.base_link,
            "joint_names_all": self.joint_names_all,
            "link_names_all": self.link_names_all,
            "joint_types_all": self.joint_types_all,
            "joint_dof": self.joint_dof, "dof_joint": self.dof_joint,
            "base_mass": repr(float(self.base_mass)), "meta": self.meta,
        }
        for k in self._ARRAYS:
            a = np.asarray(getattr(self, k))
            if k in self._INT:
                out[k] = a.astype(int).tolist()
            else:  # repr keeps all 17 significant digits
                out[k] = np.vectorize(repr)(a.astype(float)).tolist() if a.size else []
        return json.dumps(out, indent=1)

    @staticmethod
    def from_json(text: str) -> "FlatModel":
        o = json.loads(text)
        kw = {}
        for k in FlatModel._ARRAYS:
            if k not in o:          # files written before the field existed
                continue
            if k in FlatModel._INT:
                kw[k] = np.array(o[k], np.int32)
            else:
                kw[k] = np.array(o[k], dtype=object).astype(np.float64) if len(o[k]) else np.zeros(0)
        n = len(kw["parent"])
        kw["R0"] = kw["R0"].reshape(n, 3, 3)
        for k, w in (("sphere_pos", 3), ("sphere_half", 3)):
            if k in kw:
                kw[k] = kw[k].reshape(-1, w)
        return FlatModel(
            name=o["name"], base_link=o["base_link"],
            joint_names_all=o["joint_names_all"], link_names_all=o["link_names_all"],
            joint_types_all=o["joint_types_all"], joint_dof=o["joint_dof"],
            dof_joint=o["dof_joint"], base_mass=float(o["base_mass"]),
            meta=o.get("meta", {}), **kw)
