"""Differential fuzzing over LAUNCH SHAPES: the robot models stepped as whole launches of 1 .. 40 000 envs (partial last
waves of a round, one resident round exactly, one env more, many rounds: the wave-priority flip and its `flip_from`, the
per-env strides of every output), every env compared with the oracle as in fuzz_common.py.
usage (GPU box): python tools/fuzz_batch.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from fuzz_common import Totals, run_case
from humanoid_gym_amd.robots import load_robot

SIZES = [1, 2, 3, 63, 64, 65, 255, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8193, 12289, 20000, 40000]


def run(n_cases=18, seed=1):
    rng = np.random.default_rng(seed)
    tot = Totals()
    for case in range(n_cases):
        robot = str(rng.choice(["dancer", "nao", "pepper", "romeo", "nao_hands", "romeo_hands"]))
        n_envs = SIZES[case % len(SIZES)] if n_cases >= len(SIZES) else int(rng.choice(SIZES))
        if robot in ("romeo_hands", "nao_hands") and n_envs > 12289:
            n_envs = 12289          # the oracle side of these takes a minute otherwise
        model, cfg = load_robot(robot)
        model.lower, model.upper = model.lower.astype(np.float32).astype(np.float64), model.upper.astype(np.float32).astype(np.float64)
        cfg.enable_self = bool(rng.random() < 0.7)
        cfg.max_contacts = int(rng.choice([4, 8, 12]))
        sub = int(rng.integers(1, 3))
        idx = np.asarray(model.act_joint)
        lo, hi = np.maximum(model.lower[idx], -1.5), np.minimum(model.upper[idx], 1.5)
        tol = max(2e-4, 1e-4 * sub)
        tag = f"case {case}: {robot} x {n_envs} envs, self {cfg.enable_self}, max_contacts {cfg.max_contacts}, substeps {sub}"
        print(tag, flush=True)
        run_case(tot, tag, model, cfg, sub, rng, lo, hi, len(idx), tol=tol, n_envs=n_envs, n_steps=6 if n_envs > 2048 else 16)
    print("fuzz batch:", tot.line() + "; no failure")
    tot.print_ill()
    return tot


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 18, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
