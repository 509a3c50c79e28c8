"""dancer-v0 on the MI355X simulator.

Mirrors /root/reference/humanoid_gym/envs/dancer_env.py: same public
attributes (``joint_names``, ``joint2Index``, ``link_names``, ``link2Index``,
``lower_limits``, ``upper_limits``, ``init_angles``, ``step_counter``,
``action_space``, ``observation_space``), same ``step(action,
custom_reward=None) -> (observation, reward, done, info)`` and ``reset() ->
None`` contract; the PyBullet calls inside ``step`` are replaced by one
``hg_step`` launch over all envs.
"""
from __future__ import annotations

import numpy as np

from .. import spaces
from ..robots import load_robot
from .base import BatchedHumanoidEnv


class DancerEnv(BatchedHumanoidEnv):
    def __init__(self, n_envs: int = 1, device=None, urdf_path=None, self_collision: bool = True,
                 n_substeps: int = 1, auto_reset: bool = False, **cfg_overrides):
        """self_collision=True is the reference's only mode (loadURDF flags, dancer_env.py:60): `collision` and
        the contact pairs then include link-link contacts (dancer_env.py:33-39); False is an opt-out."""
        model, cfg = load_robot("dancer", urdf_path)
        for k, v in cfg_overrides.items():
            setattr(cfg, k, v)
        cfg.enable_self = bool(self_collision)
        super().__init__(model, cfg, n_envs, device, n_substeps=n_substeps, auto_reset=auto_reset)
        # dancer_env.py:66-84 -- maps built from getJointInfo order
        self.joint_names = list(model.joint_names_all)
        self.joint2Index = dict(model.joint2Index)
        self.link_names = list(model.link_names_all)
        self.link2Index = dict(model.link2Index)
        n = len(self.joint_names)
        self.lower_limits = [-np.pi] * n
        self.upper_limits = [np.pi] * n
        self.init_angles = [0] * n
        self.init_angles[7], self.init_angles[13] = -0.05, 0.05       # dancer_env.py:84 (never applied there either)
        self.action_space = spaces.Box(np.array([-1] * n, dtype=np.float32), np.array([1] * n, dtype=np.float32))
        self.observation_space = spaces.Box(np.array([-1] * n, dtype=np.float32), np.array([1] * n, dtype=np.float32))
        self.step_counter = 0

    def step(self, action, custom_reward=None):
        # dancer_env.py:19-20: targets for every joint in joint_names order, one stepSimulation
        a = self._actions_tensor(action, len(self.joint_names))
        obs, reward, done, collision = self.sim.step(a, n_substeps=self.n_substeps)
        self.step_counter += 1
        if custom_reward is None:                                     # dancer_env.py:43-46
            r = reward if self.batched else 0
            d = (done != 0) if self.batched else False
        else:                                                         # dancer_env.py:47-49
            r, d = custom_reward(jointStates=self.joint_states(), linkStates=self.link_states(),
                                 collision=(collision != 0) if self.batched else bool(collision[0].item()),
                                 step_counter=self.step_counter)
        info = {"collision": (collision != 0) if self.batched else bool(collision[0].item())}
        if self.batched:
            r, d = self._finish_step(r, d)
        return self._out(obs), r, d, info

    # -- what the reference hands to custom_reward ------------------------------
    def joint_states(self):
        """name -> (position, velocity, reactionForces, appliedJointMotorTorque) like getJointState
        (dancer_env.py:24).  reactionForces is six zeros, as in the reference, which never enables the
        joint force/torque sensors."""
        q, qd = self.sim.joint_states()
        tau = self.sim.joint_torques()
        out = {}
        for name in self.joint_names:
            k = self.model.dof_of(name)
            if self.batched:
                out[name] = (q[:, k], qd[:, k], (0.0,) * 6, tau[:, k])
            else:
                out[name] = (float(q[0, k]), float(qd[0, k]), (0.0,) * 6, float(tau[0, k]))
        return out

    def link_states(self):
        """name -> getLinkState 6-tuple (COM pos, COM quat, local inertial pos/quat, frame pos, frame quat)."""
        cp, cq, fp, fq = self.sim.link_states()
        out = {}
        for name in self.link_names:
            b = self.model.dof_of(self.joint_names[self.link2Index[name]]) + 1
            # local inertial frame = internal frame expressed in the URDF link frame (inverse of frame_pos/quat)
            lq = self.model.frame_quat[b] * np.array([-1, -1, -1, 1.0])
            from ..urdf import quat_to_mat
            lp = -quat_to_mat(lq) @ self.model.frame_pos[b]
            if self.batched:
                out[name] = (cp[:, b], cq[:, b], tuple(lp), tuple(lq), fp[:, b], fq[:, b])
            else:
                out[name] = (tuple(cp[0, b].tolist()), tuple(cq[0, b].tolist()), tuple(lp), tuple(lq),
                             tuple(fp[0, b].tolist()), tuple(fq[0, b].tolist()))
        return out

    def contact_pairs(self):
        """(link index A, link index B) of every contact of the last step; ground = -2 (dancer_env.py:36-37)."""
        pairs, counts, _ = self.sim.contact_pairs()
        return pairs, counts

    def reset(self):                                                  # dancer_env.py:55-57
        self.sim.reset()
        self.step_counter = 0
