"""Workload definitions of the reference's example drivers.

* random control (/root/reference/examples/random_control_nao.py:9):
  ``action_space.sample()`` each step, i.e. uniform in [lower_j, upper_j].
* h5 replay (/root/reference/examples/h5_control_nao.py:6-12,22 and
  h5_control_pepper.py): file ``inference.h5``, group ``group1``, datasets
  ``l_joint_angle[T,k]`` and ``r_joint_angle[T,k]``; the action row is
  ``l[t] ++ [1] ++ r[t] ++ [1]`` (hand fully open).  ``inference.h5`` is not in
  the reference tree (``*.h5`` is git-ignored there), so a synthetic trajectory
  with the same layout stands in; it is stored as ``.npz`` with the dataset
  paths as keys when h5py is absent.

The NAO replay script feeds a 12-wide Pepper-style row to NaoEnv.step, whose
positional mapping is onto 22 names (SURVEY.md 2.2 "Mismatch"); this build
defines the NAO replay through ``joints=``: the 10 arm joints plus LHand, RHand.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

L_ARM = ["LShoulderPitch", "LShoulderRoll", "LElbowYaw", "LElbowRoll", "LWristYaw"]
R_ARM = ["RShoulderPitch", "RShoulderRoll", "RElbowYaw", "RElbowRoll", "RWristYaw"]


def nao_replay_joints() -> List[str]:
    return L_ARM + ["LHand"] + R_ARM + ["RHand"]


def synthetic_h5_trajectory(model, T: int = 240, seed: int = 1234) -> Tuple[np.ndarray, np.ndarray]:
    """l_joint_angle[T,5], r_joint_angle[T,5] float32: per joint
    mid + 0.4*range*sin(2*pi*f*t/T + phi), f in {1,2,3}, phi from default_rng(seed)."""
    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None]
    out = []
    for names in (L_ARM, R_ARM):
        idx = [model.dof_of(n) for n in names]
        lo, hi = model.lower[idx], model.upper[idx]
        f = rng.integers(1, 4, size=5)
        phi = rng.uniform(0, 2 * np.pi, size=5)
        out.append((0.5 * (lo + hi) + 0.4 * (hi - lo) * np.sin(2 * np.pi * f * t / T + phi)).astype(np.float32))
    return out[0], out[1]


def pack_rows(l: np.ndarray, r: np.ndarray) -> np.ndarray:
    """h5_control_nao.py:22: l[t] ++ [1] ++ r[t] ++ [1]."""
    one = np.ones((l.shape[0], 1), l.dtype)
    return np.concatenate([l, one, r, one], axis=1)


def save_trajectory(path: str, l: np.ndarray, r: np.ndarray) -> None:
    if path.endswith(".h5"):
        import h5py  # only where available
        with h5py.File(path, "w") as hf:
            g = hf.create_group("group1")
            g.create_dataset("l_joint_angle", data=l)
            g.create_dataset("r_joint_angle", data=r)
    else:
        np.savez(path, **{"group1/l_joint_angle": l, "group1/r_joint_angle": r})


def load_trajectory(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Reads ``group1/l_joint_angle`` and ``group1/r_joint_angle`` from .h5 (needs h5py) or .npz."""
    if path.endswith(".h5"):
        import h5py
        with h5py.File(path, "r") as hf:
            g = hf.get("group1")
            return np.asarray(g.get("l_joint_angle")), np.asarray(g.get("r_joint_angle"))
    z = np.load(path, allow_pickle=False)
    return z["group1/l_joint_angle"], z["group1/r_joint_angle"]


def random_actions(model, rng: np.random.Generator, n_envs: int) -> np.ndarray:
    """action_space.sample() for Box(lower, upper): uniform per controlled joint."""
    idx = np.asarray(model.act_joint)
    return rng.uniform(model.lower[idx], model.upper[idx], (n_envs, len(idx))).astype(np.float32)
