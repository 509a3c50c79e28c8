"""Replay of a joint-angle trajectory on NAO or Pepper -- what the reference's examples/h5_control_nao.py does
(/root/reference/examples/h5_control_nao.py:6-22: group1/l_joint_angle, group1/r_joint_angle; one action row per frame
= l ++ [1] ++ r ++ [1], the hands fully open), for one env or for thousands, each starting at its own frame.

    python examples/h5_control.py nao                      # synthetic trajectory (no inference.h5 ships with the reference)
    python examples/h5_control.py nao inference.h5         # a real file (.h5 needs h5py; .npz with the same dataset paths as keys works everywhere)
    python examples/h5_control.py nao --envs 4096
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humanoid_gym_amd as hg                                                           # noqa: E402
from humanoid_gym_amd.workloads import load_trajectory, nao_replay_joints, pack_rows, synthetic_h5_trajectory   # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("robot", choices=["nao", "pepper"])
    ap.add_argument("file", nargs="?")
    ap.add_argument("--envs", type=int, default=1)
    ap.add_argument("--loops", type=int, default=1)
    a = ap.parse_args(argv)
    env = hg.make(f"{a.robot}-v0", n_envs=a.envs)
    env.reset()
    if a.file:
        l, r = load_trajectory(a.file)
    else:
        l, r = synthetic_h5_trajectory(env.model, T=240, seed=1234)
    rows = pack_rows(l, r)                                  # [T, k + 1 + k + 1]
    total_frames = rows.shape[0]
    names = nao_replay_joints()      # the columns of a row: 5 left-arm joints, LHand, 5 right-arm joints, RHand (same names on Pepper)
    phase = np.arange(a.envs) % total_frames
    for t in range(a.loops * total_frames):
        action = rows[(t + phase) % total_frames] if a.envs > 1 else rows[t % total_frames]
        if a.robot == "nao":
            observation, reward, done, info = env.step(action, joints=names)     # nao_env.py:48: step(actions, joints=None)
        else:
            full = np.tile(np.asarray(env.init_angles, np.float32), (a.envs, 1)) if a.envs > 1 else np.asarray(env.init_angles, np.float32)
            idx = [env.joint_names.index(n) for n in names]
            full[..., idx] = action
            observation, reward, done, info = env.step(full)
    print(f"{a.robot}-v0: replayed {a.loops * total_frames} frames on {a.envs} env(s); driven joints: {names}")
    env.close()
    return observation


if __name__ == "__main__":
    main()
