"""Differential fuzzing of the step kernel against the oracle over random articulated trees and configs: tree size 1..62,
depth up to 12, 1..6 chains, nested side chains, 0..12 contact slots, link-link contacts on / off, joint limits on / off,
odd / even solver sweep counts, 1..3 substeps, in 30 % of the shapes link masses scaled by 0.01 .. 10 each (random boxes of sibling links often overlap at rest, so the duplicate-pair
class of fuzz_common.py is common here).  Sample classes and bars: fuzz_common.py.  Shapes the library refuses
(HG_EUNSUPPORTED / HG_EINVAL from hg_create) are counted, not failed.
usage (GPU box): python tools/fuzz_trees.py [n_cases] [seed]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from fuzz_common import Totals, run_case
from _random_urdf import write_random_tree
from humanoid_gym_amd.urdf import load_urdf
from humanoid_gym_amd.model import SimConfig
from humanoid_gym_amd._capi import HgError


def run(n_cases=60, seed=1):
    """Steps n_cases random shapes; prints the summary line and returns the totals; AssertionError on a failure."""
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp()
    tot = Totals()
    refused, dual, sizes = 0, 0, []
    for case in range(n_cases):
        n_links = int(rng.choice([1, 2, 3, 5, 8, 13, 18, 26, 31, 33, 40, 47, 55, 62]))
        n_chains = int(rng.integers(1, 7))
        max_depth = int(rng.integers(min(12, max(1, -(-n_links // n_chains))), 13))      # the generator stops early when the chains are full
        branch = tuple((int(rng.integers(0, n_chains)), int(rng.integers(1, max(2, max_depth)))) for _ in range(int(rng.integers(0, 3))))
        self_on = bool(rng.integers(0, 2))
        gen = dict(n_links=n_links, max_depth=max_depth, n_chains=n_chains, branch_at=branch, seed=int(rng.integers(1 << 30)))
        over = dict(max_contacts=int(rng.choice([0, 1, 3, 4, 6, 8, 12])), enable_self=self_on, enable_limits=bool(rng.integers(0, 2)),
                    solver_iters=int(rng.choice([1, 7, 50, 51])))
        sub = int(rng.integers(1, 4))
        tag = f"case {case}: {gen} {over} substeps {sub}"
        try:
            path, made = write_random_tree(os.path.join(tmp, f"t{case}.urdf"), **gen)
            model = load_urdf(path, base_position=(0, 0, float(rng.uniform(0.05, 0.3))), self_collision=self_on, name="tree")
        except Exception as e:          # the generator cannot build this shape (e.g. too shallow for the link count)
            print(tag, "-> generator:", str(e)[:80]); refused += 1; continue
        scaled = rng.random() < 0.3
        if scaled:                      # mass ratios of up to 1000 between neighbouring links (the generator draws 0.05 .. 0.4 kg)
            k = 10.0 ** rng.uniform(-2, 1, model.n_links)
            model.mass, model.inertia = model.mass * k, model.inertia * k[:, None]
        cfg = SimConfig()
        for k, v in over.items():
            setattr(cfg, k, v)
        try:
            # (scaled: a light link under a heavy subtree -- the articulated inertia's rank-one downdate cancels and float32 loses
            #  three digits there; such samples must still be EXPLAINED by the oracle's spread: fuzz_common.py)
            run_case(tot, tag, model, cfg, sub, rng, -1.0, 1.0, model.n_act, tol=2e-4, base_vel=0.5, urdf=path)
        except HgError as e:
            print(tag, "-> refused:", str(e)[:100]); refused += 1; continue
        dual += int(model.n_links + 3 * cfg.max_contacts > 64); sizes.append(model.n_links)
    print("fuzz trees:", tot.line() + f"; links per shape {min(sizes)}..{max(sizes)}, {dual} shapes with more than 64 rows, "
          f"{refused} refused by the generator or hg_create; no failure")
    tot.print_ill()
    tot.refused = refused
    return tot


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
