"""HIP FP32 step (through the C-ABI) against the FP64 CPU oracle, same seeded inputs.

Parity is "unpinned" with respect to PyBullet; what is checked here
is the product kernel against the oracle restatement:
  * contact-free trajectories: joint positions within 1e-4 rad over 240 steps
    (the tolerance BASELINE.json's north_star states);
  * with ground contact: bounded drift of joint positions and base height, and
    identical integer outputs (contact link pairs, counts, collision flags).
"""
import numpy as np
import pytest

from humanoid_gym_amd.robots import load_robot
from _oracle import OracleSim

pytestmark = pytest.mark.gpu


# A contact SET may differ between the FP32 kernel and the FP64 oracle only where some candidate (a sphere
# against the plane, a sphere pair) sits within FP32 rounding of the 0.02 m manifold margin, or where the 4th
# and 5th deepest point of one body are that close (the "4 per body" rule flips).  Positions are O(1 m) and
# go through up to 12 chained FP32 transforms: 2e-6 m.  Every mismatch must be explained this way
# (VERDICT r1 item 3); everything else must be exactly equal.
SLACK_TOL = 2e-6


class IntParity:
    """Book-keeping of the integer outputs (contact count, link pairs, link mask, collision flag) of one-step
    comparisons from shared states."""

    def __init__(self, slack_tol=SLACK_TOL):
        self.total = self.same = self.explained = 0
        self.worst_slack = 0.0
        self.slack_tol = slack_tol

    def pre(self, cpu):
        """call on the shared state BEFORE stepping: slack of every env's contact set.  (The reported contact set is that of
        the LAST substep: a caller that steps several substeps calls this on the oracle's state before the last one, with a
        `slack_tol` that covers the FP32 drift of the substeps before it.)"""
        self._slack = [min(cpu.contact_slack(e)) for e in range(cpu.n_envs)]

    def check(self, e, o, counts, pairs, mask=None, col=None):
        """True if env e's integer outputs are identical; otherwise the difference must be explained."""
        self.total += 1
        ok = counts[e] == o["n_contacts"][e] and np.array_equal(pairs[e], o["pairs"][e])
        if ok:
            if mask is not None:
                assert mask[e] == o["link_mask"][e]
            if col is not None:
                assert bool(col[e]) == bool(o["collision"][e])
            self.same += 1
            return True
        assert self._slack[e] < self.slack_tol, (
            f"env {e}: contact sets differ (gpu {counts[e]} {pairs[e].tolist()} vs oracle {o['n_contacts'][e]} "
            f"{o['pairs'][e].tolist()}) but the nearest candidate is {self._slack[e]:.3e} m from the margin")
        self.explained += 1
        self.worst_slack = max(self.worst_slack, self._slack[e])
        return False

    def report(self, tag):
        print(f"[int-parity] {tag}: {self.same}/{self.total} identical, {self.explained} explained "
              f"(candidate within {self.slack_tol:g} m of the margin; worst {self.worst_slack:.2e} m), 0 unexplained")
        assert self.same == self.total - self.explained
        assert self.explained <= 0.05 * self.total, (self.explained, self.total)


# One-step float comparison under CONTACT.  A fixed number of Gauss-Seidel sweeps over a degenerate row set -- several
# contact rows between one pair of links, friction rows whose bounds follow a normal impulse that sits at 0, 1e5 N m
# motors pressing interpenetrating links together -- is not a continuous function of its rounding errors: a few
# samples in 10^5 come out O(1e-3) rad away from the FP64 result in ANY FP32 formulation.  Such a sample is accepted
# only if it is EXPLAINED (as IntParity does for the integers): the oracle is asked how far its own
# answer for that very sample moves when its constraint rows carry float32-sized rounding errors (OracleSim.noise_spread:
# 256 trials, every entry of every row's impulse response, right-hand side and diagonal perturbed by 1e-6 relative), and
# the kernel's difference must lie within EXPLAIN_FACTOR times that spread, quantity by quantity.  Everything else --
# every contact-free sample, every sample whose oracle answer is steady -- must be within `tol`; explained samples are
# bounded by `bound` and capped at max(1, 0.3 %) of the samples with contacts (the worst case of the suite: a random
# 45-link tree that two interpenetrating boxes fling about, 2 of 890).
EXPLAIN_FACTOR = 6.0
EXPLAIN_TRIALS = 256
EXPLAIN_EPS = 1e-6


class FloatParity:
    def __init__(self, tol_q, tol_qd=None, tol_tau=None, bound=0.05, min_allowed=1, explain_contact_free=False, noisy_frac=0.02,
                 bound_within_own_spread=True, noisy_cap=None):
        # the fuzzers' floor under the cap of the samples within the oracle's own spread: ONE env's worth of samples per case (one env of the
        # five that flails through a whole case -- interpenetrating links, most of its samples at the velocity clamp -- is
        # noisy in every sample that is still compared: trees seed 129 case 45, 17 of the 81 compared samples of 200 stepped)
        self.noisy_cap = noisy_cap
        # the fuzzers switch this off: a sample whose ORACLE answer moves by more than the kernel differs is bounded by that
        # spread, not by `bound` (tools/fuzz_fail/r3more4_tree435: two interpenetrating links of a 40-link tree at 20 rad/s
        # under 7 sweeps; kernel - oracle 0.058 rad where the oracle spreads by 0.135 rad under 1e-8 relative noise)
        self.bound_within_own_spread = bound_within_own_spread
        self.noisy_frac = noisy_frac        # share of the samples that may differ by less than the oracle's own spread (fuzzers: 0.1,
                                            # one chaotic env of five fills a case with them)
        self.min_allowed = min_allowed      # the cap's floor: 1 in every test; the fuzzers (thousands of random configs) use 3
        # contact-free samples: within tolerance, full stop, in every test.  The fuzzers' trees (mass ratios of 1000, 25 rad/s)
        # may explain them like the others: the oracle's noise probe reaches the articulated-body pass itself
        self.explain_contact_free = explain_contact_free
        self.tol = (tol_q, tol_qd, tol_tau)
        self.bound = bound
        self.n = self.n_contact = 0
        self.worst = [0.0, 0.0, 0.0]          # over the samples within tolerance
        self.noisy = 0                         # beyond a tolerance, but by less than the oracle's own spread
        self.outliers = []                     # explained ones: (n_contacts, |dq|)
        self.spreads = []                      # their oracle spreads (q, qd, tau)
        self._ctx = None

    def pre(self, cpu, act=None, joint_sel=None, n_substeps=1, base_cmd=None):
        """call with the shared state BEFORE stepping: what an outlier's explanation re-runs the oracle on."""
        self._ctx = (cpu, cpu.state.copy(), None if act is None else np.array(act, np.float64), joint_sel, n_substeps,
                     None if base_cmd is None else np.array(base_cmd, np.float64))

    def explain(self, e):
        cpu, before, act, sel, n_sub, cmd = self._ctx
        return cpu.noise_spread(before[e], None if act is None else act[e], joint_sel=sel, n_substeps=n_sub,
                                base_cmd_row=None if cmd is None else cmd[e], trials=EXPLAIN_TRIALS, eps=EXPLAIN_EPS)

    def add(self, n_contacts, dq, dqd=None, dtau=None, env=None, spread=None):
        """env: index of the sample in the state given to pre() (needed to explain an outlier); or `spread` directly."""
        self.n += 1
        self.n_contact += 1 if n_contacts > 0 else 0
        vals = (dq, dqd, dtau)
        bad = any(v is not None and t is not None and not (v < t) for v, t in zip(vals, self.tol))
        if bad:
            assert n_contacts > 0 or self.explain_contact_free, f"contact-free sample outside the tolerance: {vals} vs {self.tol}"
            assert np.isfinite(dq) and (dq < self.bound or not self.bound_within_own_spread), f"outlier beyond the bound: |dq| = {dq}"
            if spread is None:
                assert env is not None and self._ctx is not None, f"outlier {vals} with nothing to explain it by (call pre(), pass env=)"
                spread = self.explain(env)
            sp = (max(spread["q"], spread["base"]), spread["qd"], spread["tau_rel"])
            for v, t, s_ in zip(vals, self.tol, sp):
                if v is not None and t is not None and not (v < t):
                    assert v <= EXPLAIN_FACTOR * s_, (
                        f"UNEXPLAINED outlier: kernel - oracle = {vals} (tolerances {self.tol}), {n_contacts} contacts, but the "
                        f"oracle's own answer moves only by {sp} under float32-sized row noise ({EXPLAIN_TRIALS} trials)")
            # a sample whose oracle answer spreads by MORE than the kernel differs is not a marginal explanation: the oracle is
            # the noisier of the two there.  Such samples are counted apart (bounded at 2 % of the samples), not against the cap
            if all(v is None or t is None or v < t or v <= s_ for v, t, s_ in zip(vals, self.tol, sp)):
                self.noisy += 1
            else:
                assert dq < self.bound, f"outlier beyond the bound: |dq| = {dq} (oracle's own spread {sp})"
                self.outliers.append((int(n_contacts), float(dq)))
                self.spreads.append(tuple(float(x) for x in sp))
        else:
            for k, v in enumerate(vals):
                if v is not None:
                    self.worst[k] = max(self.worst[k], float(v))

    def report(self, tag):
        allowed = max(self.min_allowed, int(np.ceil(0.003 * (self.n if self.explain_contact_free else self.n_contact))))
        print(f"[float-parity] {tag}: {self.n} samples ({self.n_contact} with contacts), worst |dq| {self.worst[0]:.2e} rad"
              f"{'' if self.tol[1] is None else f', |dqd| {self.worst[1]:.2e} rad/s'}"
              f"{'' if self.tol[2] is None else f', torque {self.worst[2]:.2e} (relative)'}; "
              f"explained outliers (n_contacts, |dq|): {self.outliers}, oracle spreads {self.spreads} (allowed {allowed}); "
              f"{self.noisy} more within the oracle's own spread")
        assert len(self.outliers) <= allowed, (self.outliers, allowed)
        assert self.noisy <= max(2, self.noisy_cap or 0, int(np.ceil(self.noisy_frac * self.n))), (self.noisy, self.n)


def _mk(name, n_envs, **cfg_kw):
    import torch
    from humanoid_gym_amd._capi import HgSim
    model, cfg = load_robot(name)
    for k, v in cfg_kw.items():
        setattr(cfg, k, v)
    return model, cfg, HgSim(model, cfg, n_envs, "cuda:0"), OracleSim(model, cfg, n_envs), torch


def test_dancer_contact_free_240_steps():
    n_envs = 8
    model, cfg, gpu, cpu, torch = _mk("dancer", n_envs, enable_ground=False, enable_self=False, gravity=(0.0, 0.0, 0.0))
    rng = np.random.default_rng(4)
    worst = 0.0
    for t in range(240):
        if t % 24 == 0:
            act = rng.uniform(-1, 1, (n_envs, model.n_act))
        o = cpu.step(act)
        gq, _, _, _ = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"))
        worst = max(worst, float(np.abs(gq.cpu().numpy() - o["q"]).max()))
    assert worst < 1e-4, worst
    st = gpu.get_state().cpu().numpy()
    assert np.abs(st[:, :3] - cpu.state[:, :3]).max() < 1e-4
    assert np.abs(st[:, 3:7] - cpu.state[:, 3:7]).max() < 1e-4


@pytest.mark.parametrize("self_on", [False, True])
def test_dancer_free_fall_then_stand(self_on):
    """self_on=True is the reference's mode (dancer_env.py:60); at q = 0 the feet spheres overlap, so two
    self-contact rows are active from the first step and the robot lands on 8 foot corners with them."""
    n_envs = 4
    model, cfg, gpu, cpu, torch = _mk("dancer", n_envs, enable_self=self_on)
    act = np.zeros((n_envs, model.n_act))
    if self_on:      # dancer_env.py:84: the reference's init_angles spread the hips
        act[:, 7], act[:, 13] = -0.05, 0.05
    ta = torch.tensor(act, dtype=torch.float32, device="cuda:0")
    for t in range(120):
        o = cpu.step(act)
        gq, _, _, gcol = gpu.step(ta)
        pairs, counts, mask = gpu.contact_pairs()
        np.testing.assert_array_equal(counts.cpu().numpy(), o["n_contacts"], err_msg=f"step {t}")
        np.testing.assert_array_equal(pairs.cpu().numpy(), o["pairs"], err_msg=f"step {t}")
        np.testing.assert_array_equal(gcol.cpu().numpy(), o["collision"])
        np.testing.assert_array_equal(mask.cpu().numpy().astype(np.uint64), o["link_mask"])
        assert np.abs(gq.cpu().numpy() - o["q"]).max() < 2e-3, t
        if t == 0:
            first_pairs = o["pairs"].copy()
    st = gpu.get_state().cpu().numpy()
    assert np.abs(st[:, 2] - cpu.state[:, 2]).max() < 2e-3      # base height
    n_ground = (o["pairs"][..., 1] == -2).sum(1)
    assert n_ground.min() == 8                                   # standing on 2 x 4 foot corners
    if self_on:
        assert (first_pairs[..., 1] >= 0).any(1).all()           # the feet touched each other at q = 0 and were pushed apart


def test_link_states_match_oracle():
    n_envs = 3
    model, cfg, gpu, cpu, torch = _mk("dancer", n_envs, enable_ground=False, enable_self=False)
    rng = np.random.default_rng(0)
    act = rng.uniform(-1, 1, (n_envs, model.n_act))
    for _ in range(10):
        cpu.step(act)
        gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"))
    ref = cpu.link_states()
    got = [x.cpu().numpy() for x in gpu.link_states()]
    for a, b in zip(got, ref):
        if a.shape[-1] == 4:   # quaternions: q and -q are the same rotation
            s = np.sign((a * b).sum(-1, keepdims=True))
            a = a * s
        assert np.abs(a - b).max() < 2e-4
    q, qd = gpu.joint_states()
    assert np.abs(q.cpu().numpy() - cpu.q).max() < 1e-4
    assert np.abs(qd.cpu().numpy() - cpu.qd).max() < 5e-3


def _replay(model):
    from humanoid_gym_amd.workloads import nao_replay_joints, synthetic_h5_trajectory, pack_rows
    l, r = synthetic_h5_trajectory(model, T=240, seed=1234)
    rows = pack_rows(l, r)
    sel = np.array([model.dof_of(n) for n in nao_replay_joints()], np.int32)
    return rows, sel


def test_nao_replay_contact_free_1s():
    """north_star tolerance: joint positions within 1e-4 rad over a 1 s contact-free replay."""
    n_envs = 8
    model, cfg, gpu, cpu, torch = _mk("nao", n_envs, enable_ground=False, enable_self=False, gravity=(0.0, 0.0, 0.0))
    rows, sel = _replay(model)
    phase = np.arange(n_envs) * 7
    worst = 0.0
    for t in range(240):
        act = rows[(t + phase) % 240]
        o = cpu.step(act, joint_sel=sel)
        gq, _, _, _ = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"), joint_sel=sel)
        worst = max(worst, float(np.abs(gq.cpu().numpy() - o["q"]).max()))
    assert worst < 1e-4, worst


def test_nao_replay_with_ground_contact():
    """With contact: bounded drift (joint angles, base pose) and identical integer outputs."""
    n_envs = 8
    model, cfg, gpu, cpu, torch = _mk("nao", n_envs)
    rows, sel = _replay(model)
    phase = np.arange(n_envs) * 11
    for t in range(240):
        act = rows[(t + phase) % 240]
        o = cpu.step(act, joint_sel=sel)
        gq, _, _, gcol = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"), joint_sel=sel)
        pairs, counts, mask = gpu.contact_pairs()
        np.testing.assert_array_equal(counts.cpu().numpy(), o["n_contacts"], err_msg=f"step {t}")
        np.testing.assert_array_equal(pairs.cpu().numpy(), o["pairs"], err_msg=f"step {t}")
        np.testing.assert_array_equal(gcol.cpu().numpy(), o["collision"])
        assert np.abs(gq.cpu().numpy() - o["q"]).max() < 5e-3, t
    st = gpu.get_state().cpu().numpy()
    assert np.abs(st[:, :3] - cpu.state[:, :3]).max() < 5e-3          # bounded CoM-frame drift
    assert np.isfinite(st).all()


@pytest.mark.parametrize("name", ["pepper", "romeo"])
def test_other_robots_random_targets(name):
    from humanoid_gym_amd.workloads import random_actions
    n_envs = 4
    model, cfg, gpu, cpu, torch = _mk(name, n_envs)
    rng = np.random.default_rng(3)
    for t in range(60):
        if t % 20 == 0:
            act = random_actions(model, rng, n_envs)
        o = cpu.step(act)
        gq, _, _, gcol = gpu.step(torch.tensor(act, device="cuda:0"))
        assert np.abs(gq.cpu().numpy() - o["q"]).max() < 5e-3, t
    _, counts, _ = gpu.contact_pairs()
    np.testing.assert_array_equal(counts.cpu().numpy(), o["n_contacts"])
    xy = gpu.base_xyyaw.cpu().numpy()
    assert np.abs(xy - o["base_xyyaw"]).max() < 5e-3


def test_reset_mask_and_substeps():
    n_envs = 6
    model, cfg, gpu, cpu, torch = _mk("dancer", n_envs)
    act = np.full((n_envs, model.n_act), 0.2)
    ta = torch.tensor(act, dtype=torch.float32, device="cuda:0")
    for _ in range(5):
        cpu.step(act, n_substeps=4)
        gpu.step(ta, n_substeps=4)
    assert np.abs(gpu.get_state().cpu().numpy() - cpu.state).max() < 2e-3
    mask = torch.tensor([1, 0, 1, 0, 0, 1], dtype=torch.uint8, device="cuda:0")
    before = gpu.get_state().cpu().numpy()
    gpu.reset(mask)
    after = gpu.get_state().cpu().numpy()
    fresh = OracleSim(model, cfg, 1).state[0]
    for e in range(n_envs):
        if mask[e]:
            assert np.abs(after[e] - fresh).max() < 1e-6
        else:
            np.testing.assert_array_equal(after[e], before[e])


def test_dancer_self_collision_rows():
    """URDF_USE_SELF_COLLISION (dancer_env.py:60): at q = 0 the two feet overlap -- the reason the
    reference nudges init_angles[7], [13] -- so self-contact rows are active from the first step."""
    n_envs = 4
    model, cfg, gpu, cpu, torch = _mk("dancer", n_envs, enable_self=True)
    rng = np.random.default_rng(5)
    act = np.zeros((n_envs, model.n_act))
    act[:, 7], act[:, 13] = 0.4, -0.4
    act[1:] += rng.uniform(-0.3, 0.3, (n_envs - 1, model.n_act))
    ta = torch.tensor(act, dtype=torch.float32, device="cuda:0")
    saw_self = False
    worst = np.zeros(n_envs)
    same = 0
    for t in range(100):
        o = cpu.step(act)
        gq, _, _, gcol = gpu.step(ta)
        pairs, counts, mask = gpu.contact_pairs()
        c, p = counts.cpu().numpy(), pairs.cpu().numpy()
        worst = np.maximum(worst, np.abs(gq.cpu().numpy() - o["q"]).max(1))
        if t < 12:      # before trajectories can separate: integer outputs must be identical
            np.testing.assert_array_equal(c, o["n_contacts"], err_msg=f"step {t}")
            np.testing.assert_array_equal(p, o["pairs"], err_msg=f"step {t}")
            np.testing.assert_array_equal(mask.cpu().numpy().astype(np.uint64), o["link_mask"])
            assert worst.max() < 1e-3
        same += int(sum(np.array_equal(p[e], o["pairs"][e]) for e in range(n_envs)))
        saw_self |= bool(((p[..., 1] >= 0) & (p[..., 0] >= 0)).any())
        if t == 0:      # the feet (links 11, 17) overlap at q = 0
            assert set(map(tuple, o["pairs"][0, :o["n_contacts"][0]].tolist())) >= {(11, 17)}
    # Free-running FP32 and FP64 trajectories of a robot standing on deeply overlapping foot spheres separate
    # (ill-conditioned contact normals), so no joint-angle bound is asserted for the free run beyond the first
    # 12 steps above: the parity statement is the one-step comparison from shared states below.
    print(f"[free run] dancer self-collision: {same}/{100 * n_envs} (env, step) samples with identical pair lists, "
          f"worst |dq| per env {np.round(worst, 4).tolist()}")
    # ... but a bound on the free run is kept: measured 392/400 identical pair lists and 0.031 rad in the
    # worst env over the 100 steps; a kernel that is wrong, not merely float32, separates at once and everywhere
    assert same >= 0.9 * 100 * n_envs and worst.max() < 0.15 and np.median(worst) < 5e-3
    assert saw_self
    assert any(b >= 0 for _, b in o["pairs"][0, :o["n_contacts"][0]].tolist())      # link-link contacts at the end too (hips, legs)
    st = gpu.get_state().cpu().numpy()
    assert np.isfinite(st).all() and np.abs(st[:, :3] - cpu.state[:, :3]).max() < 2e-2

    # The parity statement proper: one step from the SAME state (the GPU is re-seeded with the
    # oracle's state, rounded to float32 on both sides, before every step), 60 steps in a row.
    step_err, ip = 0.0, IntParity()
    for t in range(60):
        cpu.state[:] = cpu.state.astype(np.float32)
        gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
        ip.pre(cpu)
        o = cpu.step(act)
        gq, _, _, gcol = gpu.step(ta)
        pairs, counts, mask = gpu.contact_pairs()
        pairs, counts, mask = pairs.cpu().numpy(), counts.cpu().numpy(), mask.cpu().numpy().astype(np.uint64)
        for e in range(n_envs):
            if ip.check(e, o, counts, pairs, mask, gcol.cpu().numpy()):
                step_err = max(step_err, float(np.abs(gq[e].cpu().numpy() - o["q"][e]).max()))
    ip.report("dancer self-collision, one step from shared states")
    assert step_err < 5e-4, step_err   # FP32 vs FP64 through 150 PGS sweeps over ill-conditioned (overlapping-sphere) rows


@pytest.mark.gpu
@pytest.mark.parametrize("name,steps,cfg_kw", [
    ("dancer", 80, {}), ("nao", 120, {}), ("pepper", 80, {}), ("romeo", 80, {}),      # defaults: ground + self contacts
    ("dancer", 80, dict(enable_self=False)), ("romeo", 80, dict(enable_self=False)),    # ground only
    ("romeo", 80, dict(max_contacts=4)),                                                # tools/config_table.py's config-4 row
])
def test_one_step_parity_from_shared_states(name, steps, cfg_kw):
    """The parity statement that does not depend on how fast FP32 and FP64 trajectories of a falling,
    contact-rich robot separate: before every step the GPU is re-seeded with the oracle's state (both
    rounded to float32), random targets change every 12 steps, and ONE step is compared.  Integer
    outputs (contact count, link pairs, per-link mask, collision flag) must be identical except where
    a sphere sits within FP32 rounding of the contact margin; joint angles agree to 1e-4 rad."""
    n_envs = 8
    model, cfg, gpu, cpu, torch = _mk(name, n_envs, **cfg_kw)
    rng = np.random.default_rng(11)
    lo, hi = np.asarray(model.lower)[model.act_joint], np.asarray(model.upper)[model.act_joint]
    lo, hi = np.maximum(lo, -1.5), np.minimum(hi, 1.5)
    ip, fp, saw_contacts, saw_self = IntParity(), FloatParity(1e-4, 0.03), 0, 0      # qd: rad/s; q error = dt * qd error, dt = 1/240
    for t in range(steps):
        if t % 12 == 0:
            act = rng.uniform(lo, hi, (n_envs, model.n_act)).astype(np.float32).astype(np.float64)
        cpu.state[:] = cpu.state.astype(np.float32)
        gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
        ip.pre(cpu); fp.pre(cpu, act)
        o = cpu.step(act)
        gq, _, _, gcol = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"))
        pairs, counts, mask = gpu.contact_pairs()
        pairs, counts, mask = pairs.cpu().numpy(), counts.cpu().numpy(), mask.cpu().numpy().astype(np.uint64)
        gq, gqd = gq.cpu().numpy(), gpu.obs_qd.cpu().numpy()
        saw_contacts += int(o["n_contacts"].sum())
        saw_self += int((o["pairs"][..., 1] >= -1).sum())
        gcol = gcol.cpu().numpy()
        for e in range(n_envs):
            if ip.check(e, o, counts, pairs, mask, gcol):
                fp.add(o["n_contacts"][e], float(np.abs(gq[e] - o["q"][e]).max()), float(np.abs(gqd[e] - o["qd"][e]).max()), env=e)
    ip.report(f"{name} {cfg_kw}")
    fp.report(f"{name} {cfg_kw}: {saw_contacts} contact rows, {saw_self} of them link-link")
    assert saw_contacts > 0
    if cfg.enable_self and name in ("dancer", "romeo"):
        assert saw_self > 0          # BASELINE config 4 / dancer_env.py:60: self-contact rows were exercised


@pytest.mark.gpu
def test_edge_sizes_and_contact_slot_saturation():
    """n_envs = 1 and an odd n_envs; fewer contact slots than touching points (the first max_contacts
    candidates in sphere order are kept, like the oracle); the largest n_substeps the ABI accepts."""
    for n_envs in (1, 1001):
        model, cfg, gpu, cpu, torch = _mk("dancer", n_envs)
        act = np.zeros((n_envs, model.n_act))
        for t in range(3):
            o = cpu.step(act)
            gq = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"))[0]
        assert tuple(gq.shape) == (n_envs, model.n_act)
        assert np.abs(gq.cpu().numpy() - o["q"]).max() < 1e-5
        gpu.close()
    model, cfg, gpu, cpu, torch = _mk("nao", 4, max_contacts=3)
    act = np.tile(np.asarray(model.q_init)[model.act_joint], (4, 1))
    ta = torch.tensor(act, dtype=torch.float32, device="cuda:0")
    for t in range(20):
        o = cpu.step(act)
        gq = gpu.step(ta)[0]
        pairs, counts, _ = gpu.contact_pairs()
        np.testing.assert_array_equal(counts.cpu().numpy(), o["n_contacts"])
        np.testing.assert_array_equal(pairs.cpu().numpy(), o["pairs"])
    assert int(counts.max()) == 3
    assert np.abs(gq.cpu().numpy() - o["q"]).max() < 1e-4
    model, cfg, gpu, cpu, torch = _mk("dancer", 2, enable_ground=False, enable_self=False)
    act = np.full((2, model.n_act), 0.2)
    o = cpu.step(act, n_substeps=1024)
    gq = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"), n_substeps=1024)[0]
    assert np.abs(gq.cpu().numpy() - o["q"]).max() < 2e-4


@pytest.mark.gpu
def test_pepper_commanded_base_velocity():
    """hg_set_base_velocity (the three action slots pepper_env.py:53 declares).  The GPU
    follows the oracle's restatement step for step, Pepper actually drives along its heading, and
    releasing the base (NULL) restores the reference's passive base."""
    n_envs = 4
    model, cfg, gpu, cpu, torch = _mk("pepper", n_envs)
    act = np.tile(np.asarray(model.q_init)[model.act_joint], (n_envs, 1))
    ta = torch.tensor(act, dtype=torch.float32, device="cuda:0")
    cmd = np.array([[0.3, 0.0, 0.0], [0.0, 0.2, 0.0], [0.0, 0.0, 1.0], [0.25, -0.1, 0.5]])
    gpu.set_base_velocity(torch.tensor(cmd, dtype=torch.float32))
    for t in range(120):
        o = cpu.step(act, base_cmd=cmd)
        gq = gpu.step(ta)[0]
        np.testing.assert_allclose(gpu.base_xyyaw.cpu().numpy(), o["base_xyyaw"], atol=2e-3)
        assert np.abs(gq.cpu().numpy() - o["q"]).max() < 2e-3
    assert abs(o["base_xyyaw"][0, 0] - 0.3 * 0.5) < 0.03 and abs(o["base_xyyaw"][0, 1]) < 0.02     # 0.5 s at 0.3 m/s
    assert abs(o["base_xyyaw"][1, 1] - 0.2 * 0.5) < 0.03
    # turning in place: ground friction at the (non-rolling) stand-in wheels eats part of each step's yaw rate
    assert 0.3 < o["base_xyyaw"][2, 2] < 0.55 and np.hypot(*o["base_xyyaw"][2, :2]) < 0.02
    gpu.set_base_velocity(None)
    before = gpu.base_xyyaw.clone()
    for t in range(60):
        o = cpu.step(act)
        gpu.step(ta)
    np.testing.assert_allclose(gpu.base_xyyaw.cpu().numpy(), o["base_xyyaw"], atol=5e-3)
    assert float((gpu.base_xyyaw - before)[0, 0].abs()) < 0.2           # coasts to a stop on friction


_TREES = [
    # (generator arguments, config overrides): chosen so that every kernel variant of hg_sim.hip runs
    (dict(n_links=46, max_depth=12, n_chains=4, branch_at=((0, 3), (0, 5)), seed=1), dict(max_contacts=6)),   # 46 rows, nested forks
    (dict(n_links=38, max_depth=10, n_chains=5, branch_at=((1, 2),), seed=2), dict(max_contacts=8)),          # 33..40 rows
    (dict(n_links=12, max_depth=12, n_chains=1, seed=3), dict(max_contacts=8)),                               # depth-12 chain
    (dict(n_links=28, max_depth=6, n_chains=5, seed=4), dict(max_contacts=12)),                               # 28 rows, 12 contacts
    (dict(n_links=27, max_depth=7, n_chains=3, branch_at=((0, 2), (1, 2), (2, 2)), seed=5), dict(max_contacts=8)),  # 3 forks in a row: slot reuse
    (dict(n_links=20, max_depth=6, n_chains=4, seed=6), dict(max_contacts=8, enable_self=True)),               # self-collision pairs
    (dict(n_links=45, max_depth=12, n_chains=4, branch_at=((0, 3),), seed=7), dict(max_contacts=6)),          # odd n x 18 contact columns: Jc tail quad
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(_TREES)))
def test_random_trees_one_step_parity(case, tmp_path):
    """Random articulated trees (tests/_random_urdf.py) through URDF loader, oracle and GPU: one step
    from a shared state, every step of a 0.6 s fall onto the ground with random motor targets."""
    import torch
    from _random_urdf import write_random_tree
    from humanoid_gym_amd.urdf import load_urdf
    from humanoid_gym_amd.model import SimConfig
    from humanoid_gym_amd._capi import HgSim
    gen, over = _TREES[case]
    path, made = write_random_tree(str(tmp_path / "tree.urdf"), **gen)
    model = load_urdf(path, base_position=(0, 0, 0.25), self_collision=bool(over.get("enable_self")), name="tree")
    assert model.n_links == gen["n_links"]
    cfg = SimConfig()
    for k, v in over.items():
        setattr(cfg, k, v)
    n_envs = 6
    gpu, cpu = HgSim(model, cfg, n_envs, "cuda:0"), OracleSim(model, cfg, n_envs)
    rng = np.random.default_rng(100 + case)
    cpu.state[:, 7:13] = rng.uniform(-0.5, 0.5, (n_envs, 6))            # tumbling start, different per env
    ip, fp, contacts = IntParity(), FloatParity(2e-4), 0
    for t in range(150):
        if t % 15 == 0:
            act = rng.uniform(-1.0, 1.0, (n_envs, model.n_act)).astype(np.float32).astype(np.float64)
        cpu.state[:] = cpu.state.astype(np.float32)
        gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
        ip.pre(cpu); fp.pre(cpu, act)
        o = cpu.step(act)
        gq = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"))[0].cpu().numpy()
        pairs, counts, mask = gpu.contact_pairs()
        pairs, counts, mask = pairs.cpu().numpy(), counts.cpu().numpy(), mask.cpu().numpy().astype(np.uint64)
        contacts += int(o["n_contacts"].sum())
        for e in range(n_envs):
            if ip.check(e, o, counts, pairs, mask):
                fp.add(o["n_contacts"][e], float(np.abs(gq[e] - o["q"][e]).max()), env=e)
    ip.report(f"random tree {case}")
    fp.report(f"random tree {case}")
    assert contacts > 0
    assert np.isfinite(cpu.state).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_violate", [("nao", 1), ("nao", 3), ("nao", 4), ("nao", 7), ("pepper", 2), ("romeo", 3)])
def test_joint_limit_rows(name, n_violate):
    """btMultiBodyJointLimitConstraint rows: with the motors switched off (max force 0), joints are placed
    beyond their bounds (alternating upper / lower) with velocities pointing further out.  One step
    from that state must match the oracle, and the rows must turn the violated joints around to the
    Baumgarte velocity -penetration * erp / dt; with the rows switched off they keep drifting out.
    n_violate <= 4 runs on the spare-lane path of the NAO/Pepper kernels, 7 and Romeo (no spare
    lanes) on the generic one."""
    import torch
    from humanoid_gym_amd._capi import HgSim
    n_envs = 4
    model, cfg = load_robot(name)
    model.max_force = np.zeros(model.n_links)
    cfg.enable_ground, cfg.enable_self, cfg.gravity = False, False, (0.0, 0.0, 0.0)
    lim = np.flatnonzero(model.limited)
    rng = np.random.default_rng(3)
    n = model.n_links
    for rows_on in (True, False):
        cfg.enable_limits = rows_on
        gpu, cpu = HgSim(model, cfg, n_envs, "cuda:0"), OracleSim(model, cfg, n_envs)
        for trial in range(6 if rows_on else 1):
            cpu.state[:] = OracleSim(model, cfg, n_envs).state
            pick = rng.choice(lim, size=n_violate, replace=False)
            sgn = np.array([-1.0 if (k + trial) % 2 == 0 else 1.0 for k in range(n_violate)])   # inward direction
            for k, j in enumerate(pick):
                cpu.q[:, j] = (model.upper[j] + 0.05 + 0.02 * k) if sgn[k] < 0 else (model.lower[j] - 0.05 - 0.02 * k)
                cpu.qd[:, j] = -0.5 * sgn[k]
            cpu.state[:] = cpu.state.astype(np.float32)
            gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
            pen = np.array([0.05 + 0.02 * k for k in range(n_violate)])
            for t in range(3):
                cpu.step(None)
                gpu.step(None)
                st = gpu.get_state().cpu().numpy()
                assert np.abs(st[:, 13:13 + n] - cpu.q).max() < 1e-4
                assert np.abs(st[:, 13 + n:13 + 2 * n] - cpu.qd).max() < 2e-2
                if t == 0:
                    v_in = sgn[None, :] * cpu.qd[:, pick]
                    if rows_on:      # erp 0.2, dt 1/240
                        assert (v_in > 0.5 * pen * 0.2 * 240).all(), (name, pick, v_in)
                    else:
                        assert (v_in < 0).all(), (name, pick, v_in)
                cpu.state[:] = cpu.state.astype(np.float32)
                gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
        gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dancer", "nao", "pepper"])
def test_applied_motor_torque(name):
    """hg_get_joint_torques = getJointState()[3] (dancer_env.py:24): the motor row's impulse / dt of the
    last physics step, against the oracle's, from shared states with random targets; zero after reset."""
    n_envs = 8
    model, cfg, gpu, cpu, torch = _mk(name, n_envs)
    assert float(gpu.joint_torques().abs().max()) == 0.0
    rng = np.random.default_rng(21)
    lo, hi = np.asarray(model.lower)[model.act_joint], np.asarray(model.upper)[model.act_joint]
    lo, hi = np.maximum(lo, -1.5), np.minimum(hi, 1.5)
    worst, scale_seen, ip = 0.0, 0.0, IntParity()
    for t in range(40):
        if t % 8 == 0:
            act = rng.uniform(lo, hi, (n_envs, model.n_act)).astype(np.float32).astype(np.float64)
        cpu.state[:] = cpu.state.astype(np.float32)
        gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
        ip.pre(cpu)
        o = cpu.step(act)
        gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"))
        pairs, counts, _ = gpu.contact_pairs()
        pairs, counts = pairs.cpu().numpy(), counts.cpu().numpy()
        tau = gpu.joint_torques().cpu().numpy()
        for e in range(n_envs):
            if not ip.check(e, o, counts, pairs):
                continue                  # explained: a candidate within FP32 rounding of the margin, different row sets
            scale = max(1.0, float(np.abs(o["motor_torque"][e]).max()))
            scale_seen = max(scale_seen, scale)
            worst = max(worst, float(np.abs(tau[e] - o["motor_torque"][e]).max()) / scale)
    assert scale_seen > 1.0                                    # the motors did something
    assert worst < 5e-3, worst                                 # relative to the env's largest torque
    fmax = np.asarray(model.max_force)
    assert (np.abs(tau) <= fmax[None, :] * (1 + 1e-5) + 1e-6).all()      # |impulse| <= max_force * dt
    gpu.reset()
    assert float(gpu.joint_torques().abs().max()) == 0.0


@pytest.mark.gpu
def test_store_torques_off_changes_nothing_but_the_getter():
    """hg_config.store_torques = 0 (what the qibullet-style envs and the bench use: nothing there reads
    getJointState()[3]): the step is bit-identical, hg_get_joint_torques fails loudly instead of returning stale zeros."""
    import torch
    import humanoid_gym_amd as hg
    import dataclasses
    from humanoid_gym_amd._capi import HgError, HgSim
    n_envs = 5
    model, cfg = load_robot("nao_hands")
    on = HgSim(model, cfg, n_envs, "cuda:0")
    off = HgSim(model, dataclasses.replace(cfg, store_torques=False), n_envs, "cuda:0")
    rng = np.random.default_rng(2)
    for t in range(12):
        act = torch.tensor(rng.uniform(-0.5, 0.5, (n_envs, model.n_act)).astype(np.float32), device="cuda:0")
        a, b = on.step(act), off.step(act)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(on.get_state(), off.get_state())
    assert float(on.joint_torques().abs().max()) > 0.0
    with pytest.raises(HgError, match="store_torques"):
        off.joint_torques()
    on.close(); off.close()
    env = hg.make("nao-v0", n_envs=2)
    assert env.sim.cfg.store_torques is False and hg.make("dancer-v0", n_envs=2).sim.cfg.store_torques is True


@pytest.mark.gpu
def test_store_contacts_off_and_no_velocity_buffer_change_nothing_else():
    """hg_config.store_contacts = 0 and a NULL obs_qd_dev (what NaoEnv and the bench hand over: the reference's NAO
    observation is the joint positions, nao_env.py:73-79, and nothing there calls getContactPoints): every other output
    and the state are bit-identical, the `collision` flag included; hg_get_contact_pairs fails loudly."""
    import torch
    import humanoid_gym_amd as hg
    import dataclasses
    from humanoid_gym_amd._capi import HgError, HgSim
    n_envs = 6
    model, cfg = load_robot("nao_hands")
    on = HgSim(model, cfg, n_envs, "cuda:0")
    off = HgSim(model, dataclasses.replace(cfg, store_contacts=False), n_envs, "cuda:0")
    off.obs_qd = None
    rng = np.random.default_rng(3)
    seen = 0
    for t in range(12):
        act = torch.tensor(rng.uniform(-0.5, 0.5, (n_envs, model.n_act)).astype(np.float32), device="cuda:0")
        a, b = on.step(act), off.step(act)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        assert torch.equal(on.base_xyyaw, off.base_xyyaw)
        seen += int(off.collision.sum())
    assert seen > 0                                            # the flag is live without the lists
    assert torch.equal(on.get_state(), off.get_state())
    assert int(on.contact_pairs()[1].sum()) > 0
    with pytest.raises(HgError, match="store_contacts"):
        off.contact_pairs()
    on.close(); off.close()
    nao, pepper = hg.make("nao-v0", n_envs=2), hg.make("pepper-v0", n_envs=2)
    assert nao.sim.cfg.store_contacts is False and nao.sim.obs_qd is None
    assert pepper.sim.cfg.store_contacts is False and pepper.sim.obs_qd is not None      # pepper_env.py:67: velocities are observed
    obs, _, _, _ = nao.step(np.asarray(nao.init_angles, np.float32))
    assert np.isfinite(np.asarray(obs.cpu() if hasattr(obs, "cpu") else obs)).all()
    dancer = hg.make("dancer-v0", n_envs=2)
    assert dancer.sim.cfg.store_contacts is True
    dancer.step(np.zeros(18, np.float32)); dancer.contact_pairs()
