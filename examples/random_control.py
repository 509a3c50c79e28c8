"""Random control of one of the four simulated robots -- what the reference's examples/random_control_<robot>.py
do (`env.action_space.sample()` every step, /root/reference/examples/random_control_nao.py:9), on the MI355X
library instead of PyBullet, for one env with the reference's return types or for thousands at once.

    python examples/random_control.py nao                  # 1 env: lists / floats / bools like the reference
    python examples/random_control.py dancer --envs 4096   # batched: device tensors with a leading env axis
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humanoid_gym_amd as hg      # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("robot", choices=["nao", "pepper", "romeo", "dancer"])
    ap.add_argument("--envs", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    env = hg.make(f"{a.robot}-v0", n_envs=a.envs)
    env.reset()
    rng = np.random.default_rng(a.seed)
    lo, hi = np.asarray(env.action_space.low), np.asarray(env.action_space.high)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        actions = rng.uniform(lo, hi, (a.envs, lo.size) if a.envs > 1 else lo.size).astype(np.float32)
        observation, reward, done, info = env.step(actions)
    dt = time.perf_counter() - t0
    print(f"{a.robot}-v0: {a.envs} env(s) x {a.steps} steps in {dt:.2f} s ({a.envs * a.steps / dt:.0f} env-steps/s incl. host-side sampling); "
          f"last done = {done if a.envs == 1 else int(done.sum())}")
    env.close()
    return observation


if __name__ == "__main__":
    main()
