"""Differential fuzzing of the step kernel against the oracle over random articulated trees and configs: tree size 1..62,
depth up to 12, 1..6 chains, nested side chains, 0..12 contact slots, link-link contacts on / off, joint limits on / off,
odd / even solver sweep counts, 1..3 substeps.  One step from shared float32 states, same bars as the GPU tests
(IntParity: integer outputs identical or explained by margin slack; FloatParity: 2e-4 rad, rare degenerate contact sets
counted and bounded).  Samples that start with a velocity coordinate within 10 % of the +-100 clamp (two interpenetrating
boxes flinging a tree about: the clamp is a discontinuity, and such an env stays there for many steps) are counted apart
and not compared; so are samples with SEVERAL contacts between one pair of links (random boxes of sibling links often
overlap at rest: the Delassus matrix is singular, how the impulse splits between the duplicates is decided by rounding,
and the friction bounds follow the split: 2 % of such samples differ by 1e-4 .. 7e-3 rad in any FP32 formulation, see
profiles/round2_d_one_step_outliers.txt).  Shapes the library refuses (HG_EUNSUPPORTED / HG_EINVAL from hg_create) are counted, not failed.
usage (GPU box): python tools/fuzz_trees.py [n_cases] [seed]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(1, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from _random_urdf import write_random_tree
from _oracle import OracleSim
from test_gpu_parity import FloatParity, IntParity
from humanoid_gym_amd.urdf import load_urdf
from humanoid_gym_amd.model import SimConfig
from humanoid_gym_amd._capi import HgError, HgSim


def run(n_cases=60, seed=1):
    """Steps n_cases random cases; prints and returns the summary line; AssertionError on a failure."""
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp()
    refused, ran, samples, outliers, contact_samples, dual, sizes, n_sat, n_dup = 0, 0, 0, 0, 0, 0, [], 0, 0
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
        cfg = SimConfig()
        for k, v in over.items():
            setattr(cfg, k, v)
        n_envs = 5
        try:
            gpu = HgSim(model, cfg, n_envs, "cuda:0")
        except HgError as e:
            print(tag, "-> refused:", str(e)[:100]); refused += 1; continue
        cpu = OracleSim(model, cfg, n_envs)
        cpu.state[:, 7:13] = rng.uniform(-0.5, 0.5, (n_envs, 6))
        ip, fp = IntParity(), FloatParity(2e-4)
        for t in range(40):
            if t % 10 == 0:
                act = rng.uniform(-1.0, 1.0, (n_envs, model.n_act)).astype(np.float32).astype(np.float64)
            cpu.state[:] = cpu.state.astype(np.float32)
            gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
            ip.pre(cpu)
            vmax = cfg.max_coord_velocity
            saturated = (np.abs(cpu.state[:, 7:13]).max(1) > 0.9 * vmax) | (np.abs(cpu.qd).max(1) > 0.9 * vmax)
            o = cpu.step(act, n_substeps=sub)
            gq = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"), n_substeps=sub)[0].cpu().numpy()
            pairs, counts, mask = gpu.contact_pairs()
            pairs, counts, mask = pairs.cpu().numpy(), counts.cpu().numpy(), mask.cpu().numpy().astype(np.uint64)
            blown = ~np.isfinite(cpu.state).all(1) | (np.abs(np.nan_to_num(cpu.state, nan=1e30)).max(1) > 1e6)      # e.g. ONE solver sweep on a 62-link tree
            for e in range(n_envs):
                nce = int(o["n_contacts"][e])
                ll = [(int(a), int(b)) for a, b in o["pairs"][e][:nce] if b >= 0]
                if saturated[e] or blown[e]:
                    n_sat += 1
                elif len(ll) != len(set(ll)):
                    n_dup += 1
                elif ip.check(e, o, counts, pairs, mask):
                    fp.add(o["n_contacts"][e], float(np.abs(gq[e] - o["q"][e]).max()))
            if blown.any():
                break
        try:
            ip.report(tag); fp.report(tag)
        except AssertionError as e:
            print("FAILED", tag, e); raise AssertionError("fuzz failure (details printed above): " + tag)
        ran += 1; samples += fp.n; outliers += len(fp.outliers); contact_samples += fp.n_contact
        dual += int(model.n_links + 3 * cfg.max_contacts > 64); sizes.append(model.n_links)
        gpu.close()
    print(f"fuzz: {ran} shapes stepped ({samples} samples, {contact_samples} of them with contacts, {outliers} counted outliers, {n_sat} samples at the velocity clamp (or with a diverged oracle) and {n_dup} with duplicate link-link contacts not compared; "
          f"links per shape {min(sizes)}..{max(sizes)}, {dual} shapes with more than 64 rows), {refused} refused by the generator or hg_create; no failure")
    return dict(cases=ran, samples=samples, contact_samples=contact_samples, outliers=outliers, refused=refused)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
