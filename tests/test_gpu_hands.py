"""Robots at their real size: the NAO-like model with the 16 finger joints of its hands (42 dofs, 66
constraint rows with 8 contacts) and the Romeo-like one with 24 (61 dofs, 85 rows) -- romeo_env.py:24-26
shows that the finger / thumb joints are simulated dofs of the reference's robots.  More rows than the 64
lanes of a wavefront: the kernel carries two rows per lane (hg_step.hip, DUAL).  Same bars as the other
robots: 1e-4 rad over the 1 s contact-free replay, one step from shared states with identical integer
outputs, joint-limit rows, applied torques."""
import numpy as np
import pytest

from humanoid_gym_amd.robots import load_robot
from _oracle import OracleSim
from test_gpu_parity import FloatParity, IntParity

pytestmark = pytest.mark.gpu


def _mk(name, n_envs, **cfg_kw):
    import torch
    from humanoid_gym_amd._capi import HgSim
    model, cfg = load_robot(name)
    for k, v in cfg_kw.items():
        setattr(cfg, k, v)
    return model, cfg, HgSim(model, cfg, n_envs, "cuda:0"), OracleSim(model, cfg, n_envs), torch


def _replay_with_fingers(model):
    """h5_control_nao rows (l ++ [1] ++ r ++ [1]) on the 10 arm joints + LHand / RHand, plus one column per
    finger joint that mimics a hand (what qibullet's setAngles / the env layer adds): 12 + 16 columns."""
    from humanoid_gym_amd.workloads import nao_replay_joints, synthetic_h5_trajectory, pack_rows
    l, r = synthetic_h5_trajectory(model, T=240, seed=1234)
    rows = pack_rows(l, r)
    names = nao_replay_joints()
    sel = [model.dof_of(n) for n in names]
    cols = [rows]
    for child, (parent, mult, off) in model.meta["mimic"].items():
        if parent in names:
            sel.append(model.dof_of(child))
            cols.append((mult * rows[:, names.index(parent)] + off)[:, None])
    return np.concatenate(cols, axis=1).astype(np.float32), np.array(sel, np.int32)


def test_nao_with_hands_is_accepted_and_replays_contact_free_1e4():
    """hg_create accepts a 42-dof NAO-like with 8 contact slots, and the 1e-4 rad replay
    statement holds on it."""
    n_envs = 8
    model, cfg, gpu, cpu, torch = _mk("nao_hands", n_envs, enable_ground=False, enable_self=False, gravity=(0.0, 0.0, 0.0))
    assert model.n_links == 42 and cfg.max_contacts == 8 and model.n_links + 3 * cfg.max_contacts == 66
    rows, sel = _replay_with_fingers(model)
    assert rows.shape == (240, 28)
    phase = np.arange(n_envs) * 7
    worst = 0.0
    for t in range(240):
        act = rows[(t + phase) % 240]
        o = cpu.step(act, joint_sel=sel)
        gq, _, _, _ = gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"), joint_sel=sel)
        worst = max(worst, float(np.abs(gq.cpu().numpy() - o["q"]).max()))
    st = gpu.get_state().cpu().numpy()
    n = model.n_links
    worst_all = float(np.abs(st[:, 13:13 + n] - cpu.q).max())          # every dof, fingers included
    print(f"[hands] nao_hands contact-free replay: worst |dq| controlled {worst:.2e}, all dofs {worst_all:.2e} rad")
    assert worst < 1e-4 and worst_all < 1e-4, (worst, worst_all)
    lf = model.dof_of("LFinger12")
    assert np.abs(cpu.q[:, lf] - 0.95).max() < 0.05                     # the fingers followed LHand = 1 (open)


@pytest.mark.parametrize("name,steps,cfg_kw", [
    ("nao_hands", 100, {}), ("romeo_hands", 80, {}),                    # ground + link-link contacts, 66 / 85 rows
    ("romeo_hands", 60, dict(enable_self=False)),
    ("nao_hands", 60, dict(max_contacts=5, solver_iters=51)),           # fewer contacts than slots, odd sweep count
])
def test_hands_one_step_parity_from_shared_states(name, steps, cfg_kw):
    n_envs = 8
    model, cfg, gpu, cpu, torch = _mk(name, n_envs, **cfg_kw)
    rng = np.random.default_rng(12)
    n = model.n_links
    # every dof gets a target (fingers too): sel = all dofs
    sel = np.arange(n, dtype=np.int32)
    lo, hi = np.maximum(model.lower, -1.5), np.minimum(model.upper, 1.5)
    ip, fp, saw_contacts, saw_self = IntParity(), FloatParity(1e-4, 0.03, 5e-3), 0, 0
    for t in range(steps):
        if t % 12 == 0:
            act = rng.uniform(lo, hi, (n_envs, n)).astype(np.float32).astype(np.float64)
        cpu.state[:] = cpu.state.astype(np.float32)
        gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
        ip.pre(cpu); fp.pre(cpu, act, joint_sel=sel)
        o = cpu.step(act, joint_sel=sel)
        gpu.step(torch.tensor(act, dtype=torch.float32, device="cuda:0"), joint_sel=sel)
        pairs, counts, mask = gpu.contact_pairs()
        pairs, counts, mask = pairs.cpu().numpy(), counts.cpu().numpy(), mask.cpu().numpy().astype(np.uint64)
        st = gpu.get_state().cpu().numpy()
        tau = gpu.joint_torques().cpu().numpy()
        saw_contacts += int(o["n_contacts"].sum())
        saw_self += int((o["pairs"][..., 1] >= -1).sum())
        gcol = gpu.collision.cpu().numpy()
        for e in range(n_envs):
            if ip.check(e, o, counts, pairs, mask, gcol):
                scale = max(1.0, float(np.abs(o["motor_torque"][e]).max()))
                before = len(fp.outliers)
                fp.add(o["n_contacts"][e], float(np.abs(st[e, 13:13 + n] - cpu.q[e]).max()),
                       float(np.abs(st[e, 13 + n:13 + 2 * n] - cpu.qd[e]).max()),
                       float(np.abs(tau[e] - o["motor_torque"][e]).max()) / scale, env=e)
                if len(fp.outliers) == before:
                    assert np.abs(st[e, :3] - cpu.state[e, :3]).max() < 1e-4
    ip.report(f"{name} {cfg_kw}")
    fp.report(f"{name} {cfg_kw}: {saw_contacts} contact rows ({saw_self} link-link)")
    assert saw_contacts > 0


@pytest.mark.parametrize("name,n_violate", [("nao_hands", 1), ("nao_hands", 6), ("romeo_hands", 3), ("romeo_hands", 9)])
def test_hands_joint_limit_rows(name, n_violate):
    """Joint-limit rows on the two-rows-per-lane variants (always the run-time loop over violated joints):
    motors off, joints beyond their bounds moving further out; one step must match the oracle and turn them around."""
    import torch
    from humanoid_gym_amd._capi import HgSim
    n_envs = 4
    model, cfg = load_robot(name)
    model.max_force = np.zeros(model.n_links)
    cfg.enable_ground, cfg.enable_self, cfg.gravity = False, False, (0.0, 0.0, 0.0)
    lim = np.flatnonzero(model.limited)
    rng = np.random.default_rng(4)
    n = model.n_links
    gpu, cpu = HgSim(model, cfg, n_envs, "cuda:0"), OracleSim(model, cfg, n_envs)
    for trial in range(5):
        cpu.state[:] = OracleSim(model, cfg, n_envs).state
        pick = rng.choice(lim, size=n_violate, replace=False)
        sgn = np.array([-1.0 if (k + trial) % 2 == 0 else 1.0 for k in range(n_violate)])
        for k, j in enumerate(pick):
            cpu.q[:, j] = (model.upper[j] + 0.05 + 0.01 * k) if sgn[k] < 0 else (model.lower[j] - 0.05 - 0.01 * k)
            cpu.qd[:, j] = -0.5 * sgn[k]
        cpu.state[:] = cpu.state.astype(np.float32)
        gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
        for t in range(3):
            cpu.step(None)
            gpu.step(None)
            st = gpu.get_state().cpu().numpy()
            assert np.abs(st[:, 13:13 + n] - cpu.q).max() < 1e-4
            assert np.abs(st[:, 13 + n:13 + 2 * n] - cpu.qd).max() < 2e-2
            if t == 0:
                assert ((sgn[None, :] * cpu.qd[:, pick]) > 0).all()       # turned around
            cpu.state[:] = cpu.state.astype(np.float32)
            gpu.set_state(torch.tensor(cpu.state, dtype=torch.float32))
    gpu.close()


@pytest.mark.parametrize("gen,over,steps", [
    (dict(n_links=50, max_depth=10, n_chains=6, branch_at=((0, 2), (3, 4)), seed=21), dict(max_contacts=8), 150),               # 74 rows
    (dict(n_links=62, max_depth=12, n_chains=6, branch_at=((1, 3),), seed=22), dict(max_contacts=8), 150),                      # the largest tree: 86 rows
    # 69 rows with link-link pairs.  Late in the run one env of this tree is flung about at 25-65 rad/s by two
    # interpenetrating boxes and single steps then differ by 3e-4 .. 4e-2 rad in every kernel variant: the kind of
    # sample FloatParity counts as an outlier (test_gpu_parity.py).
    (dict(n_links=45, max_depth=9, n_chains=5, seed=23), dict(max_contacts=8, enable_self=True), 150),
    (dict(n_links=40, max_depth=8, n_chains=6, seed=24), dict(max_contacts=12), 150),                                           # catch-all variant: 40 + 36 rows
])
def test_large_random_trees_one_step_parity(gen, over, steps, tmp_path):
    import torch
    from _random_urdf import write_random_tree
    from humanoid_gym_amd.urdf import load_urdf
    from humanoid_gym_amd.model import SimConfig
    from humanoid_gym_amd._capi import HgSim
    path, made = write_random_tree(str(tmp_path / "tree.urdf"), **gen)
    model = load_urdf(path, base_position=(0, 0, 0.25), self_collision=bool(over.get("enable_self")), name="tree")
    assert model.n_links == gen["n_links"]
    cfg = SimConfig()
    for k, v in over.items():
        setattr(cfg, k, v)
    n_envs = 6
    gpu, cpu = HgSim(model, cfg, n_envs, "cuda:0"), OracleSim(model, cfg, n_envs)
    rng = np.random.default_rng(200 + gen["seed"])
    cpu.state[:, 7:13] = rng.uniform(-0.5, 0.5, (n_envs, 6))
    ip, fp, contacts = IntParity(), FloatParity(2e-4), 0
    for t in range(steps):
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
    ip.report(f"large tree {gen['n_links']} links")
    fp.report(f"large tree {gen['n_links']} links")
    assert contacts > 0
    assert np.isfinite(cpu.state).all()


def test_hands_envs_and_full_size_invariants():
    """env surface (hands=True keeps the reference's controlled joint lists), 4096-env determinism and
    position independence on the 42-dof NAO."""
    import torch
    import humanoid_gym_amd as hg
    nao = hg.make("nao-v0", n_envs=4, hands=True)
    assert len(nao.joint_names) == 22 and nao.model.n_links == 42
    obs, r, d, info = nao.step(torch.tensor(np.tile(nao.init_angles, (4, 1)), dtype=torch.float32, device="cuda:0"))
    assert tuple(obs.shape) == (4, 22)
    nao.step(torch.full((4, 2), 0.8, device="cuda:0"), joints=["LHand", "RHand"])        # fingers ride along (mimic)
    n = nao.model.n_links
    tgt = nao.sim.get_state()[:, 13 + 2 * n:].cpu().numpy()
    assert abs(tgt[0, nao.model.dof_of("LFinger11")] - 0.95 * 0.8) < 1e-6
    # the column plan of a joint list is built once; a follower that is named itself takes its own column
    nao.step(torch.full((4, 2), 0.5, device="cuda:0"), joints=["LHand", "RHand"])
    assert len(nao._plans) == 2                                                             # all 22 joints, (LHand, RHand)
    nao.step(torch.tensor([[0.5, 0.1]] * 4, device="cuda:0"), joints=["LHand", "LFinger11"])
    tgt = nao.sim.get_state()[:, 13 + 2 * n:].cpu().numpy()
    assert abs(tgt[0, nao.model.dof_of("LFinger11")] - 0.1) < 1e-6 and abs(tgt[0, nao.model.dof_of("LFinger12")] - 0.95 * 0.5) < 1e-6
    nao.close()
    rom = hg.make("romeo-v0", n_envs=2, hands=True)
    assert rom.model.n_links == 61 and len(rom.joint_names) == 37 and not any("Finger" in j or "Thumb" in j for j in rom.joint_names)
    o, *_ = rom.step(torch.tensor(np.tile(rom.init_angles, (2, 1)), dtype=torch.float32, device="cuda:0"))
    assert tuple(o["anglesPosition"].shape) == (2, 37)
    rom.close()
    from humanoid_gym_amd._capi import HgSim
    model, cfg = load_robot("nao_hands")
    n_envs = 4096
    sim = HgSim(model, cfg, n_envs, "cuda:0")
    g = torch.Generator(device="cpu").manual_seed(6)
    lo = torch.tensor(np.asarray(model.lower)[model.act_joint], dtype=torch.float32)
    hi = torch.tensor(np.asarray(model.upper)[model.act_joint], dtype=torch.float32)
    base = lo + (hi - lo) * torch.rand(16, model.n_act, generator=g)
    act = base[torch.arange(n_envs) % 16].to("cuda:0").contiguous()
    for t in range(6):
        sim.step(act)
    s0 = sim.get_state().clone()
    sim.step(act); s1 = sim.get_state().clone()
    sim.set_state(s0); sim.step(act); s2 = sim.get_state()
    assert bool(torch.isfinite(s1).all()) and bool((s1 == s2).all())
    grouped = s1.view(n_envs // 16, 16, -1)
    assert bool((grouped == grouped[0:1]).all())
    sim.close()


@pytest.mark.parametrize("robot,variant,cfg_kw", [
    ("nao", 2, {}), ("nao", 3, dict(max_contacts=12)), ("nao", 6, dict(max_contacts=12)), ("nao", 7, {}), ("nao", 8, {}),
    ("nao", 9, dict(max_contacts=6)), ("nao", 10, {}), ("nao", 11, {}), ("nao", 12, {}), ("nao", 13, dict(max_contacts=12)),
    ("pepper", 5, {}), ("pepper", 6, {}), ("pepper", 10, {}), ("pepper", 11, {}), ("dancer", 3, {}), ("dancer", 13, {}),
])
def test_every_kernel_instantiation_agrees_with_the_oracle(robot, variant, cfg_kw, monkeypatch):
    """hg_sim.hip instantiates 14 shapes of the step kernel; the four robots and the two hands models pick 0, 1, 4, 7,
    10 and 12 by themselves.  HG_VARIANT (a diagnostics override read by hg_create) forces the others -- and the
    overflow-row shape 10 and the two-rows-per-lane shapes 11-13 onto robots that normally run one row per lane -- so
    that every compiled shape is held to the same one-step bar."""
    import torch
    from humanoid_gym_amd._capi import HgSim
    monkeypatch.setenv("HG_VARIANT", str(variant))
    model, cfg = load_robot(robot)
    for k, v in cfg_kw.items():
        setattr(cfg, k, v)
    n_envs = 6
    gpu, cpu = HgSim(model, cfg, n_envs, "cuda:0"), OracleSim(model, cfg, n_envs)
    monkeypatch.delenv("HG_VARIANT")
    rng = np.random.default_rng(40 + variant)
    idx = np.asarray(model.act_joint)
    lo, hi = np.maximum(model.lower[idx], -1.2), np.minimum(model.upper[idx], 1.2)
    ip, fp, contacts = IntParity(), FloatParity(2e-4), 0      # 2e-4 like the random trees: Dancer's 1e5 N m motors on the two-rows-per-lane shapes reach 1.3e-4
    for t in range(50):
        if t % 10 == 0:
            act = rng.uniform(lo, hi, (n_envs, len(idx))).astype(np.float32).astype(np.float64)
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
    ip.report(f"{robot} on variant {variant}")
    fp.report(f"{robot} on variant {variant}")
    assert contacts > 0
    gpu.close()
