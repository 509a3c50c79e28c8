"""nao-v0, pepper-v0, romeo-v0 on the MI355X simulator.

Mirror /root/reference/humanoid_gym/envs/{nao,pepper,romeo}_env.py: joint name
lists, limit / init-angle attributes, ``step`` signatures and observation
layouts.  The robot models are the stand-ins of ``humanoid_gym_amd.robots``
unless ``urdf_path`` points at SoftBank's real files.

Step definition: the reference never calls ``stepSimulation`` for these envs
(physics runs asynchronously at wall-clock rate under qibullet's GUI mode), so
an env step has no deterministic length there.  Here one env step = write
targets, ``n_substeps`` physics steps of 1/240 s (default 1), read back.
"""
from __future__ import annotations

import numpy as np

from .. import spaces
from ..robots import NAO_JOINTS, PEPPER_JOINTS, load_robot
from .base import BatchedHumanoidEnv


class _SoftbankEnv(BatchedHumanoidEnv):
    ROBOT = ""

    def __init__(self, n_envs: int = 1, device=None, urdf_path=None, n_substeps: int = 1,
                 auto_reset: bool = False, hands=None, **cfg_overrides):
        """hands (nao, romeo; default True): the model with the finger / thumb joints of the hands as simulated dofs, as
        the robots that the reference steps have them -- romeo_env.py:24-26 filters `Finger` / `Thumb` out of the
        CONTROLLED list only (42 / 61 dofs).  They follow LHand / RHand through their <mimic> tags; the controlled
        joint lists are unchanged.  hands=False: the smaller stand-ins without finger joints (26 / 37 dofs)."""
        if hands is None:
            hands = self.ROBOT in ("nao", "romeo") and urdf_path is None
        model, cfg = load_robot(self.ROBOT, urdf_path, hands=hands)
        # qibullet's robots are read through getAnglesPosition / getAnglesVelocity (nao_env.py:73, pepper_env.py:64): no
        # caller of these envs reads getJointState()[3], so the step does not store the motor torques (4 n bytes per env
        # and physics step less written); store_torques=True as an override brings sim.joint_torques() back
        cfg.store_torques = False
        # ... nor getContactPoints: the link pairs / counts / link masks (108 B per env) stay unstored as well
        # (store_contacts=True brings sim.contact_pairs() back; the per-env `collision` flag is written either way)
        cfg.store_contacts = False
        for k, v in cfg_overrides.items():
            setattr(cfg, k, v)
        super().__init__(model, cfg, n_envs, device, n_substeps, auto_reset)
        self.joint_names = [model.dof_names[k] for k in model.act_joint]
        idx = np.asarray(model.act_joint)
        self.lower_limits = model.lower[idx].tolist()
        self.upper_limits = model.upper[idx].tolist()
        self.init_angles = model.q_init[idx].tolist()
        self.link_names = [model.dof_link_names[k] for k in model.act_joint]     # nao_env.py:40-43
        # <mimic> joints of the URDF (the fingers of the real NAO / Pepper / Romeo hands) follow the joint
        # they mimic, as qibullet's setAngles does on the host: target = multiplier * parent + offset.
        self._mimic = {}
        for child, (parent, mult, off) in (model.meta.get("mimic") or {}).items():
            if child in model.dof_names and parent in model.dof_names:
                self._mimic.setdefault(parent, []).append((model.dof_of(child), float(mult), float(off)))
        self._plans = {}      # joint-name tuple -> action-column plan (_column_plan)
        n = len(self.joint_names)
        self.observation_space = spaces.Box(np.array([-1] * n, dtype=np.float32), np.array([1] * n, dtype=np.float32))

    def _column_plan(self, names):
        """(dof indices of the action columns, source column / multiplier / offset tensors or None) for a list of joint
        names: the columns given, then one column per <mimic> follower of a named joint that is not named itself."""
        sel = [self.model.dof_of(j) for j in names]
        src, mult, off = list(range(len(sel))), [1.0] * len(sel), [0.0] * len(sel)
        for c, j in enumerate(names):
            for dof, m, o in self._mimic.get(j, ()):
                if dof not in sel:
                    sel.append(dof); src.append(c); mult.append(m); off.append(o)
        if len(sel) == len(names):
            return np.asarray(sel, np.int32), None
        t = self._torch
        f = lambda v: t.tensor(v, dtype=t.float32, device=self.device)      # noqa: E731
        return np.asarray(sel, np.int32), (t.tensor(src, dtype=t.int64, device=self.device), f(mult), f(off))

    def _step_sim(self, actions, joints=None):
        names = self.joint_names if joints is None else list(joints)
        if joints is None and not self._mimic:
            a = self._actions_tensor(actions, len(names))
            return self.sim.step(a, n_substeps=self.n_substeps)
        key = tuple(names)
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 64:      # callers that keep inventing joint lists do not grow the cache for ever
                self._plans.clear()
            plan = self._plans[key] = self._column_plan(names)
        sel, cols = plan
        a = self._actions_tensor(actions, len(names))
        if cols is not None:      # one gather and one fused multiply-add: follower = multiplier * parent + offset (x 1 + 0 for the rest)
            a = self._torch.addcmul(cols[2], a.index_select(1, cols[0]), cols[1])
        return self.sim.step(a, joint_sel=sel, n_substeps=self.n_substeps)

    def _dict_obs(self):
        return {"position": self._out(self.sim.base_xyyaw) if self.batched else tuple(self.sim.base_xyyaw[0].tolist()),
                "anglesPosition": self._out(self.sim.obs_q),
                "anglesVelocity": self._out(self.sim.obs_qd)}

    def reset(self):                      # nao_env.py:83-86: respawn in the Stand posture, returns None
        self.sim.reset()


class NaoEnv(_SoftbankEnv):
    ROBOT = "nao"

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        assert self.joint_names == NAO_JOINTS                          # nao_env.py:30-31
        self.sim.obs_qd = None       # the observation is the 22 positions (nao_env.py:73-79): hg_step gets no velocity buffer
        self.action_space = spaces.Box(np.array(self.lower_limits, dtype=np.float32),
                                       np.array(self.upper_limits, dtype=np.float32))

    def step(self, actions, joints=None):                              # nao_env.py:48-81
        obs, reward, done, _ = self._step_sim(actions, joints)
        if self.batched:
            return obs, reward, done != 0, {}
        return self._out(obs), 0, False, {}


class PepperEnv(_SoftbankEnv):
    ROBOT = "pepper"

    # qibullet's limits for PepperVirtual.move [UPSTREAM-RECALL]: 0.55 m/s, 2 rad/s
    BASE_SCALE = (0.55, 0.55, 2.0)

    def __init__(self, *a, use_base_slots: bool = False, **kw):
        """use_base_slots=False reproduces the reference (the three leading action slots are ignored and
        the wheeled base is passive); True wires them to the base: slots in [-1, 1] scale to
        (vx, vy, yaw rate) = BASE_SCALE * slot (a documented extension)."""
        super().__init__(*a, **kw)
        self.use_base_slots = bool(use_base_slots)
        assert self.joint_names == PEPPER_JOINTS                       # pepper_env.py:28-39
        # pepper_env.py:53: three extra base slots are declared but never consumed by step()
        self.action_space = spaces.Box(np.array([-1] * 3 + self.lower_limits, dtype=np.float32),
                                       np.array([1] * 3 + self.upper_limits, dtype=np.float32))

    def step(self, actions):                                           # pepper_env.py:56-76
        t = self._torch
        # tensors stay on the device (no host round trip per step); lists / arrays go through numpy once
        a = actions if isinstance(actions, t.Tensor) else np.asarray(actions, dtype=np.float32)
        if a.shape[-1] == len(self.joint_names) + 3:
            if self.use_base_slots:
                cmd = t.as_tensor(a[..., :3], dtype=t.float32, device=self.device).reshape(-1, 3).clamp(-1.0, 1.0)
                cmd = cmd * t.tensor(self.BASE_SCALE, dtype=t.float32, device=self.device)
                self.sim.set_base_velocity(cmd.expand(self.n_envs, 3) if cmd.shape[0] == 1 else cmd)
            a = a[..., 3:]                 # reference behaviour: the base slots are not consumed (pepper_env.py:61)
        obs, reward, done, _ = self._step_sim(a)
        if self.batched:
            return self._dict_obs(), reward, done != 0, {}
        return self._dict_obs(), 0, False, {}


class RomeoEnv(_SoftbankEnv):
    ROBOT = "romeo"

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.action_space = spaces.Box(np.array(self.lower_limits, dtype=np.float32),
                                       np.array(self.upper_limits, dtype=np.float32))

    def step(self, actions):                                           # romeo_env.py:33-52
        obs, reward, done, _ = self._step_sim(actions)
        if self.batched:
            return self._dict_obs(), reward, done != 0, {}
        return self._dict_obs(), 0, False, {}
