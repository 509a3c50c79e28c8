"""Shared part of the two differential fuzzers (fuzz_trees.py, fuzz_robots.py): one case = one model + config + substep
count, stepped 40 times on the kernel (through the C-ABI) and the oracle from shared FP32 states, every (env, step) sample
classified.  The bars are those of the GPU tests (tests/test_gpu_parity.py: IntParity, FloatParity).

Sample classes that are NOT compared (each counted in the summary; FUZZ_STRICT=1 compares the second and third too):
  clamp   a velocity coordinate beyond 90 % of the clamp (or of 100 rad/s when the clamp is off) before or after the step,
          or a diverged oracle: the clamp is a discontinuity and a runaway env loses 1e-4 rad to the rounding of q alone;
  dup     several contacts between one pair of links (in any substep of the step): singular Delassus matrix, the split between the duplicates is
          decided by rounding (profiles/round2_d_one_step_outliers.txt);
  tie     (several substeps, limit rows on) a joint closer to a bound than the position tolerance at the start of a later
          substep: it rests on the bound under its limit row, leaves a substep at bound +- rounding noise, and whether the
          next substep builds the row again is a coin toss in any arithmetic.
Samples that ARE compared and miss a bar (joint angles beyond the tolerance; with several substeps also a contact set that
differs without a candidate at the margin -- the last substep starts from states the two sides reached by their own
arithmetic) are put to the oracle itself: 4 trials with +-1 ulp (FP32) on its input state, 4 with 1e-6 relative noise.  If its own FP64 answer moves by more
than tolerance / 20, an FP32 step -- hundreds of roundings inside -- cannot be held to the tolerance: logged as
ill-conditioned with both numbers.  Otherwise the sample counts (FloatParity's outlier allowance) or fails the run.
With ONE substep the contact set comes from the shared state and a difference the margin slack does not explain fails at once."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from _oracle import OracleSim
from test_gpu_parity import FloatParity, IntParity
from humanoid_gym_amd._capi import HgSim

STRICT = os.environ.get("FUZZ_STRICT", "0") == "1"
N_ENVS, N_STEPS = 5, 40


class Totals:
    def __init__(self):
        self.cases = self.samples = self.contact_samples = self.outliers = 0
        self.clamp = self.dup = self.tie = 0
        self.ill = []

    def line(self):
        return (f"{self.cases} cases stepped ({self.samples} samples compared, {self.contact_samples} of them with contacts, "
                f"{self.outliers} counted outliers, {len(self.ill)} ill-conditioned; not compared: {self.clamp} at the velocity clamp or "
                f"diverged, {self.dup} with duplicate link-link contacts, {self.tie} with a joint at a bound between substeps)")

    def print_ill(self):
        for rec in self.ill:
            print("  ill-conditioned: %s, %d contacts: kernel - oracle %.3g rad%s, oracle under input noise %.3g rad" % rec)


def _ulp_nudge(x64, rng):
    b32 = x64.astype(np.float32)
    moved = (b32.view(np.int32) + rng.integers(-1, 2, b32.shape).astype(np.int32)).view(np.float32)
    return np.where(b32 != 0, moved, b32)


def run_case(tot, tag, model, cfg, sub, rng, act_lo, act_hi, n_cols, sel=None, tol=2e-4, base_vel=0.3, urdf=None):
    """Steps one case; raises AssertionError (after printing what it knows) on a failure."""
    n = model.n_links
    gpu = HgSim(model, cfg, N_ENVS, "cuda:0")
    cpu, walk = OracleSim(model, cfg, N_ENVS), OracleSim(model, cfg, N_ENVS)
    cpu.state[:, 7:13] = rng.uniform(-base_vel, base_vel, (N_ENVS, 6))
    # several substeps: the slack is taken at the oracle's state before the LAST one, against 1e-4 rad of drift on a 0.5 m lever
    ip, fp = IntParity(2e-6 if sub == 1 else 5e-5), FloatParity(tol)
    lim_idx = np.flatnonzero(model.limited)
    vlim = 0.9 * (min(cfg.max_coord_velocity, 100.0) if cfg.max_coord_velocity > 0 else 100.0)
    fast = lambda st: (np.abs(st[:, 7:13]).max(1) > vlim) | (np.abs(st[:, 13 + n:13 + 2 * n]).max(1) > vlim)

    def both(state, n_sub):
        walk.state[:] = state
        g2 = HgSim(model, cfg, N_ENVS, "cuda:0")
        g2.set_state(torch.tensor(state, dtype=torch.float32))
        o2 = walk.step(act, joint_sel=sel, n_substeps=n_sub)
        q2 = g2.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"), joint_sel=sel, n_substeps=n_sub)[0].cpu().numpy()
        g2.close()
        return o2, q2

    def dump_sample(sub_dir, e, before, o, gq):
        dump = os.environ.get("FUZZ_DUMP")
        if dump:        # the sample, for tools/fuzz_replay.py
            import dataclasses, json, shutil
            dump = os.path.join(dump, sub_dir)
            os.makedirs(dump, exist_ok=True)
            np.savez(os.path.join(dump, "sample.npz"), before=before, act=act, sel=np.asarray([] if sel is None else sel, np.int32), sub=sub, env=e,
                     kernel_q=gq, oracle_q=o["q"])
            json.dump(dict(tag=tag, cfg=dataclasses.asdict(cfg), has_sel=sel is not None, urdf=urdf is not None, name=model.name), open(os.path.join(dump, "sample.json"), "w"))
            if urdf:
                shutil.copy(urdf, os.path.join(dump, "sample.urdf"))

    def fail(e, t, why, before, o, gq):
        print("FAILED", tag, f"step {t} env {e}: {why}")
        dq = np.abs(gq[e] - o["q"][e])
        j = int(dq.argmax())
        print(f"  worst column {j} (link {int(np.asarray(model.act_joint)[j])}): kernel {gq[e][j]} oracle {o['q'][e][j]}; "
              f"oracle contacts {o['pairs'][e][:int(o['n_contacts'][e])].tolist()}; |qd|max after {np.abs(cpu.qd[e]).max():.3g}")
        for s2 in range(1, sub + 1):
            o2, q2 = both(before, s2)
            print(f"  the same sample with {s2} substep(s): max |dq| {np.abs(q2[e] - o2['q'][e]).max():.3g}")
        dump_sample("", e, before, o, gq)
        raise AssertionError("fuzz failure (details printed above): " + tag)

    for t in range(N_STEPS):
        if t % 10 == 0:
            act = rng.uniform(act_lo, act_hi, (N_ENVS, n_cols)).astype(np.float32).astype(np.float64)
        cpu.state[:] = cpu.state.astype(np.float32)
        before = cpu.state.copy()
        gpu.set_state(torch.tensor(before, dtype=torch.float32))
        skip = fast(before)
        tie = np.zeros(N_ENVS, bool)
        walk.state[:] = before
        dup = np.zeros(N_ENVS, bool)
        for s2 in range(sub - 1):
            ow = walk.step(act if s2 == 0 else None, joint_sel=sel if s2 == 0 else None, n_substeps=1)
            for e in range(N_ENVS):       # the duplicate-pair class counts in EVERY substep, not only in the reported one
                ll = [(int(a), int(b)) for a, b in ow["pairs"][e][:int(ow["n_contacts"][e])] if b >= 0]
                dup[e] |= len(ll) != len(set(ll))
            if cfg.enable_limits and len(lim_idx):
                d = np.minimum(np.abs(walk.q - model.lower), np.abs(walk.q - model.upper))
                tie |= (d[:, lim_idx] < tol).any(1)
        ip.pre(walk)
        o = cpu.step(act, joint_sel=sel, n_substeps=sub)
        gq = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"), joint_sel=sel, n_substeps=sub)[0].cpu().numpy()
        pairs, counts, mask = gpu.contact_pairs()
        pairs, counts, mask = pairs.cpu().numpy(), counts.cpu().numpy(), mask.cpu().numpy().astype(np.uint64)
        blown = ~np.isfinite(cpu.state).all(1) | (np.abs(np.nan_to_num(cpu.state, nan=1e30)).max(1) > 1e6)      # e.g. ONE sweep on a 62-link tree
        skip |= blown | fast(np.nan_to_num(cpu.state, nan=1e30))
        for e in range(N_ENVS):
            nce = int(o["n_contacts"][e])
            ll = [(int(a), int(b)) for a, b in o["pairs"][e][:nce] if b >= 0]
            if skip[e]:
                tot.clamp += 1; continue
            if (dup[e] or len(ll) != len(set(ll))) and not STRICT:
                tot.dup += 1; continue
            if tie[e] and not STRICT:
                tot.tie += 1; continue
            int_why = None
            try:
                identical = ip.check(e, o, counts, pairs, mask)
            except AssertionError as err:
                identical, int_why = False, str(err)
                if sub == 1:
                    fail(e, t, int_why, before, o, gq)
            dq_e = float(np.abs(gq[e] - o["q"][e]).max())
            if int_why is not None or (identical and not dq_e < tol):
                sens = 0.0
                for trial in range(8):      # 4 x +-1 ulp, then 4 x 1e-6 relative (~16 ulp: roundings INSIDE an FP32 step add up to that)
                    walk.state[:] = before
                    walk.state[e] = _ulp_nudge(before[e], rng) if trial < 4 else before[e] * (1.0 + 1e-6 * rng.normal(size=before.shape[1]))
                    sens = max(sens, float(np.abs(walk.step(act, joint_sel=sel, n_substeps=sub)["q"][e] - o["q"][e]).max()))
                if sens > tol / 20:
                    if int_why is not None:
                        ip.total -= 1
                    tot.ill.append((tag.split(":")[0], nce, dq_e, " and another contact set" if int_why else "", sens))
                    continue
                if int_why is not None:
                    fail(e, t, int_why + f" (oracle under +-1 ulp: {sens:.3g} rad)", before, o, gq)
            if identical:
                n_out = len(fp.outliers)
                try:
                    fp.add(nce, dq_e)
                except AssertionError as err:
                    fail(e, t, str(err), before, o, gq)
                if len(fp.outliers) > n_out:
                    dump_sample(f"outlier_{tag.split(':')[0].replace(' ', '')}_{t}_{e}", e, before, o, gq)
        if blown.any():
            break
    try:
        ip.report(tag); fp.report(tag)
    except AssertionError as err:
        print("FAILED", tag, err)
        raise AssertionError("fuzz failure (details printed above): " + tag)
    tot.cases += 1; tot.samples += fp.n; tot.outliers += len(fp.outliers); tot.contact_samples += fp.n_contact
    gpu.close()
