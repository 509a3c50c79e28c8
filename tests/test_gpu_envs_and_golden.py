"""GPU: the HIP path against the committed golden trajectories, the env surface (single-env
reference semantics and batched), N-replica stats, and full-size invariants."""
import os

import numpy as np
import pytest

import humanoid_gym_amd as hg
from humanoid_gym_amd.robots import load_robot
from _oracle import OracleSim

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _sim(name, n_envs, **kw):
    import torch
    from humanoid_gym_amd._capi import HgSim
    model, cfg = load_robot(name)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return model, cfg, HgSim(model, cfg, n_envs, "cuda:0"), torch


def test_golden_nao_replay_free_1e4():
    z = np.load(os.path.join(HERE, "golden", "oracle_nao_replay_free.npz"))
    model, cfg, sim, torch = _sim("nao", 2, enable_ground=False, enable_self=False, gravity=(0.0, 0.0, 0.0))
    worst = 0.0
    for t in range(240):
        act = torch.tensor(z["rows"][(t + np.array([0, 37])) % 240], device="cuda:0")
        q, _, _, _ = sim.step(act, joint_sel=z["sel"])
        worst = max(worst, float(np.abs(q.cpu().numpy() - z["q"][t]).max()))
    assert worst < 1e-4, worst                   # north_star: 1e-4 rad over a 1 s contact-free replay


def test_golden_nao_replay_ground_and_dancer_random():
    z = np.load(os.path.join(HERE, "golden", "oracle_nao_replay_ground.npz"))
    model, cfg, sim, torch = _sim("nao", 2)
    for t in range(240):
        act = torch.tensor(z["rows"][(t + np.array([0, 37])) % 240], device="cuda:0")
        q, _, _, _ = sim.step(act, joint_sel=z["sel"])
        _, counts, _ = sim.contact_pairs()
        np.testing.assert_array_equal(counts.cpu().numpy(), z["n_contacts"][t])
        assert np.abs(q.cpu().numpy() - z["q"][t]).max() < 5e-3
    # Dancer under random targets, landing on its hull vertices with link-link contacts: ONE step from each of the
    # oracle's stored (float32) states.  Integer outputs identical (or explained by the margin slack of that state),
    # joint angles within 1e-4 rad (FloatParity: see test_gpu_parity.py).
    from test_gpu_parity import FloatParity, IntParity
    z = np.load(os.path.join(HERE, "golden", "oracle_dancer_random.npz"))
    model, cfg, sim, torch = _sim("dancer", 2)
    cpu = OracleSim(model, cfg, 2)         # only asked for the margin slack of the stored states; never stepped
    T = z["actions"].shape[0]
    ip, fp = IntParity(), FloatParity(1e-4)
    for t in range(T):
        sim.set_state(torch.tensor(z["state"][t]))
        cpu.state[:] = z["state"][t]
        ip.pre(cpu); fp.pre(cpu, z["actions"][t])
        q = sim.step(torch.tensor(z["actions"][t], device="cuda:0"))[0].cpu().numpy()
        pairs, counts, mask = sim.contact_pairs()
        pairs, counts, mask = pairs.cpu().numpy(), counts.cpu().numpy(), mask.cpu().numpy().astype(np.uint64)
        o = dict(n_contacts=z["n_contacts"][t], pairs=z["pairs"][t], link_mask=z["link_mask"][t])
        for e in range(2):
            if ip.check(e, o, counts, pairs, mask):
                fp.add(o["n_contacts"][e], float(np.abs(q[e] - z["q"][t][e]).max()), env=e)
    ip.report("golden dancer")
    fp.report("golden dancer")
    assert z["n_contacts"].max() >= 4 and (z["pairs"][..., 1] >= -1).any()      # the fixture does hold ground and link-link contacts


def test_dancer_env_single_env_reference_semantics():
    env = hg.make("dancer-v0")
    assert len(env.joint_names) == 18 and env.joint2Index["joint_body_head"] == 0
    assert env.link2Index["foot2_right"] == 17 and env.init_angles[7] == -0.05 and env.init_angles[13] == 0.05
    assert env.lower_limits == [-np.pi] * 18 and env.action_space.shape == (18,)
    assert env.reset() is None and env.step_counter == 0
    obs, reward, done, info = env.step(env.action_space.sample())
    assert isinstance(obs, list) and len(obs) == 18 and reward == 0 and done is False
    # reference flags (dancer_env.py:60): link-link contacts count, and at q = 0 the feet already touch each other
    assert set(info) == {"collision"} and isinstance(info["collision"], bool) and env.step_counter == 1
    seen = {}

    def custom(jointStates, linkStates, collision, step_counter):        # dancer_env.py:49
        seen.update(js=jointStates, ls=linkStates, col=collision, sc=step_counter)
        return 1.5, step_counter >= 40

    for _ in range(60):
        obs, reward, done, info = env.step([0.0] * 18, custom_reward=custom)
        if done:
            break
    assert reward == 1.5 and done is True and seen["sc"] == 40 and env.step_counter == 40
    assert set(seen["js"]) == set(env.joint_names) and len(seen["js"]["joint_leg_left"]) == 4
    assert set(seen["ls"]) == set(env.link_names) and len(seen["ls"]["foot2_left"]) == 6
    assert info["collision"] is True and seen["col"] is True              # standing on the ground by now
    pairs, counts = env.contact_pairs()
    p0 = pairs[0, :counts[0].item()].cpu().numpy()
    assert int((p0[:, 1] == -2).sum()) == 8                                # 2 x 4 foot corners on the ground
    assert int((p0[:, 1] >= 0).sum()) >= 1                                 # and the foot-foot contact (dancer_env.py:36-37)
    env.close()
    off = hg.make("dancer-v0", self_collision=False)                       # opt-out: ground contacts only
    for _ in range(40):
        off.step([0.0] * 18)
    pairs, counts = off.contact_pairs()
    assert counts[0].item() == 8 and bool((pairs[0, :8, 1] == -2).all())
    off.close()


def test_nao_pepper_romeo_env_surfaces():
    import torch
    nao = hg.make("nao-v0", n_envs=16)
    assert nao.joint_names[0] == "LShoulderPitch" and len(nao.joint_names) == 22 and len(nao.link_names) == 22
    a = torch.tensor(np.tile(nao.init_angles, (16, 1)), dtype=torch.float32, device="cuda:0")
    obs, reward, done, info = nao.step(a)
    assert tuple(obs.shape) == (16, 22) and info == {} and not bool(done.any())
    obs2, *_ = nao.step(torch.zeros(16, 2, device="cuda:0"), joints=["HeadYaw", "HeadPitch"])   # nao_env.py:48
    assert nao.reset() is None
    nao1 = hg.make("nao-v0")
    o, r, d, i = nao1.step(np.array(nao1.init_angles))
    assert isinstance(o, list) and len(o) == 22 and r == 0 and d is False and i == {}
    with pytest.raises(ValueError):
        nao1.step([0.0] * 5)
    pep = hg.make("pepper-v0")
    assert pep.action_space.shape == (15,)                                 # pepper_env.py:53
    o, r, d, i = pep.step(pep.action_space.sample())                       # 15-D sample: base slots ignored
    assert set(o) == {"position", "anglesPosition", "anglesVelocity"} and len(o["position"]) == 3
    assert len(o["anglesPosition"]) == 12 and len(o["anglesVelocity"]) == 12
    rom = hg.make("romeo-v0", n_envs=4)
    o, r, d, i = rom.step(torch.tensor(np.tile(rom.init_angles, (4, 1)), dtype=torch.float32, device="cuda:0"))
    assert tuple(o["anglesPosition"].shape) == (4, 37) and tuple(o["position"].shape) == (4, 3)
    # the robots are simulated at the size the reference simulates them: finger / thumb joints are dofs of the model
    # (romeo_env.py:24-26 keeps them out of the CONTROLLED list only); hands=False is the opt-out
    assert nao.model.n_links == 42 and rom.model.n_links == 61 and pep.model.n_links == 20
    assert all("Finger" not in j and "Thumb" not in j for j in rom.joint_names)
    small = hg.make("nao-v0", n_envs=2, hands=False)
    assert small.model.n_links == 26 and small.joint_names == nao.joint_names
    for e in (nao, nao1, pep, rom, small):
        e.close()


def test_full_size_invariants_4096_envs():
    """At BASELINE.json's size: env independence (identical envs stay bit-identical, a perturbed env
    does not disturb its neighbours), permutation equivariance, finite state."""
    import torch
    from humanoid_gym_amd.workloads import nao_replay_joints, pack_rows, synthetic_h5_trajectory
    model, cfg, sim, torch = _sim("nao", 4096)
    l, r = synthetic_h5_trajectory(model)
    rows = torch.tensor(pack_rows(l, r), device="cuda:0")
    sel = np.array([model.dof_of(n) for n in nao_replay_joints()], np.int32)
    phase = torch.arange(4096, device="cuda:0") % 8            # 8 distinct phases -> 512 copies of each
    for t in range(40):
        sim.step(rows[(t + phase) % 240].contiguous(), joint_sel=sel)
    st = sim.get_state()
    assert bool(torch.isfinite(st).all())
    grouped = st.view(512, 8, -1)
    assert bool((grouped == grouped[0:1]).all())                # same inputs -> bit-identical states
    _, counts, _ = sim.contact_pairs()
    assert int(counts.min()) >= 4 and int(counts.max()) <= cfg.max_contacts
    stats = torch.zeros(8, device="cuda:0")
    sim.stats_accumulate(stats)
    assert stats[0].item() == 4096 and stats[3].item() == float((sim.collision != 0).sum().item())


def test_invalid_arguments_are_rejected():
    model, cfg, sim, torch = _sim("dancer", 4)
    from humanoid_gym_amd._capi import HgError
    with pytest.raises(HgError):
        sim.step(torch.zeros(4, 18, device="cuda:0"), joint_sel=[0, 0])         # duplicate joint
    with pytest.raises(HgError):
        sim.step(torch.zeros(4, 1, device="cuda:0"), joint_sel=[99])            # out of range
    with pytest.raises(HgError):
        sim.step(torch.zeros(4, 17, device="cuda:0"))                           # wrong width
    with pytest.raises(HgError):
        sim.step(torch.zeros(4, 18, device="cuda:0"), n_substeps=0)
    bad = sim.get_state(); bad[2, 20] = float("nan"); sim.set_state(bad)
    _, _, done, _ = sim.step(torch.zeros(4, 18, device="cuda:0"))
    assert done.cpu().tolist() == [0, 0, 1, 0]                                  # env 2 reset in place
    assert bool(torch.isfinite(sim.get_state()).all())


def test_batched_custom_reward_and_auto_reset():
    """custom_reward on device tensors, masked auto-reset without host sync."""
    import torch
    env = hg.make("dancer-v0", n_envs=64, auto_reset=True)

    def custom(jointStates, linkStates, collision, step_counter):
        z = linkStates["body_head"][0][:, 2]                       # COM height of the head link
        reward = z - 0.3
        done = torch.arange(64, device=z.device) % 4 == (step_counter % 4)   # a quarter of the envs each step
        return reward, done

    fresh = env.sim.get_state()[0].clone()
    for t in range(1, 9):
        obs, reward, done, info = env.step(torch.zeros(64, 18, device="cuda:0"), custom_reward=custom)
        assert tuple(obs.shape) == (64, 18) and tuple(reward.shape) == (64,) and done.dtype == torch.bool
        st = env.sim.get_state()
        assert bool((st[done] == fresh).all())                     # just-finished envs are back at the reset pose
        assert bool((env.episode_length[done] == 0).all()) and bool((env.episode_length[~done] > 0).all())
    assert bool((env.last_episode_length > 0).all()) and bool((env.last_episode_length <= 4).all())
    assert info["collision"].dtype == torch.bool
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_envs,n_substeps", [("pepper", 16384, 1), ("romeo", 8192, 1), ("dancer", 65536, 4),
                                                    ("nao_hands", 4096, 1), ("romeo_hands", 8192, 1)])
def test_full_size_invariants_other_configs(name, n_envs, n_substeps):
    """BASELINE.json configs 3-5 at their per-GPU sizes (and configs 2 / 4 on the models with finger joints, the size
    the reference simulates), through size-independent properties: two
    launches from the same state give bit-identical results (no race, no dependence on scheduling),
    envs with equal inputs stay bit-identical wherever they sit in the batch, everything stays finite."""
    model, cfg, sim, torch = _sim(name, n_envs)
    g = torch.Generator(device="cpu").manual_seed(5)
    lo = torch.tensor(np.maximum(np.asarray(model.lower)[model.act_joint], -1.0), dtype=torch.float32)
    hi = torch.tensor(np.minimum(np.asarray(model.upper)[model.act_joint], 1.0), dtype=torch.float32)
    base = lo + (hi - lo) * torch.rand(16, model.n_act, generator=g)          # 16 distinct action rows
    idx = torch.arange(n_envs) % 16
    act = base[idx].to("cuda:0").contiguous()
    for t in range(6):
        sim.step(act, n_substeps=n_substeps)
    s0 = sim.get_state().clone()
    sim.step(act, n_substeps=n_substeps)
    s1 = sim.get_state().clone()
    sim.set_state(s0)
    sim.step(act, n_substeps=n_substeps)
    s2 = sim.get_state()
    assert bool(torch.isfinite(s1).all())
    assert bool((s1 == s2).all())                                             # deterministic
    grouped = s1.view(n_envs // 16, 16, -1)
    assert bool((grouped == grouped[0:1]).all())                              # position in the batch is irrelevant
    _, counts, _ = sim.contact_pairs()
    assert int(counts.max()) <= cfg.max_contacts
    sim.close()


@pytest.mark.gpu
def test_mimic_joints_follow_their_parent(tmp_path):
    """with a URDF that has <mimic> finger joints, commanding LHand/RHand also commands
    the fingers (target = multiplier * hand + offset), through step(actions, joints=)."""
    import torch
    import humanoid_gym_amd as hg
    from _mimic_urdf import write
    env = hg.make("nao-v0", n_envs=3, urdf_path=write(str(tmp_path / "nao_fingers.urdf")))
    m = env.model
    n, lf, rf = m.n_links, m.dof_of("LFinger11"), m.dof_of("RFinger11")
    tgt = lambda: env.sim.get_state()[:, 13 + 2 * n:].cpu().numpy()
    obs, *_ = env.step(torch.tensor(np.tile(env.init_angles, (3, 1)), dtype=torch.float32, device="cuda:0"))
    assert tuple(obs.shape) == (3, 22)           # the 22 NAO joints do not include the hands (nao_env.py:30-31)
    a = torch.tensor([[0.2, 0.4], [0.5, 0.4], [0.8, 0.4]], device="cuda:0")
    env.step(a, joints=["LHand", "RHand"])       # h5_control_nao.py:22 drives the hands this way
    np.testing.assert_allclose(tgt()[:, lf], 0.9 * np.array([0.2, 0.5, 0.8]) + 0.05, atol=1e-6)
    np.testing.assert_allclose(tgt()[:, rf], 0.9 * 0.4 + 0.05, atol=1e-6)
    env.step(torch.full((3, 1), 1.0, device="cuda:0"), joints=["RHand"])          # nao_env.py:48
    np.testing.assert_allclose(tgt()[:, rf], 0.95, atol=1e-6)
    np.testing.assert_allclose(tgt()[:, lf], 0.9 * np.array([0.2, 0.5, 0.8]) + 0.05, atol=1e-6)   # untouched
    for _ in range(60):
        env.step(torch.full((3, 1), 1.0, device="cuda:0"), joints=["RHand"])
    q = env.sim.get_state()[:, 13:13 + n].cpu().numpy()
    assert np.abs(q[:, rf] - 0.95).max() < 0.05                                   # the finger got there
    env.close()


@pytest.mark.gpu
def test_fused_statistics_equal_per_step_accumulation():
    """hg_stats_flush (running per-env counters kept by the step kernel) gives the same sums as calling
    hg_stats_accumulate after every step, is cleared by a flush and is not touched by hg_reset."""
    model, cfg, sim, torch = _sim("nao", 300)
    per_step = torch.zeros(8, device="cuda:0")
    fused = torch.zeros(8, device="cuda:0")
    bad = sim.get_state(); bad[7, 20] = float("nan"); sim.set_state(bad)      # one env raises done once
    for t in range(25):
        sim.step(None)
        sim.stats_accumulate(per_step)
        if t == 11:
            sim.stats_flush(fused)
            sim.reset()
    sim.stats_flush(fused)
    assert per_step[0].item() == 25 * 300 and per_step[2].item() == 1.0
    assert torch.equal(per_step[:4], fused[:4]), (per_step, fused)
    again = torch.zeros(8, device="cuda:0")
    sim.stats_flush(again)
    assert float(again.abs().sum()) == 0.0
    from humanoid_gym_amd.stats import EpisodeStats
    st = EpisodeStats(sim, reduce_every=4)
    for t in range(10):
        sim.step(None)
        st.update(t)
    tot = st.flush().tolist()
    assert tot[0] == 10 * 300 and tot[3] == 10 * float((sim.collision != 0).sum().item())


@pytest.mark.gpu
def test_pepper_base_slots_extension():
    """pepper-v0 with use_base_slots=True consumes the three leading action slots (pepper_env.py:53);
    the default ignores them like the reference."""
    import torch
    import humanoid_gym_amd as hg
    for use, lo, hi in ((False, -0.02, 0.02), (True, 0.15, 0.35)):
        env = hg.make("pepper-v0", n_envs=2, use_base_slots=use)
        a = np.concatenate([np.tile([0.5, 0.0, 0.0], (2, 1)), np.tile(env.init_angles, (2, 1))], axis=1).astype(np.float32)
        for _ in range(240):
            obs, *_ = env.step(a)
        x = float(obs["position"][0, 0])
        assert lo < x < hi, (use, x)                  # 1 s at 0.5 * 0.55 m/s = 0.275 m, or nothing
        env.close()


@pytest.mark.gpu
def test_random_control_soak_stays_finite():
    """random_control_*.py for 600 steps on 1024 envs per robot: with Bullet's velocity-coordinate clamp
    (hg_config.max_coord_velocity) no env ever needs the non-finite reset, Dancer's 1e5 N m motors
    included; without it (clamp off) Dancer does blow up, which is what the guard is for."""
    from humanoid_gym_amd.workloads import random_actions
    for name, sub, clamp, expect_resets in (("dancer", 4, 100.0, False), ("nao", 1, 100.0, False), ("dancer", 4, 0.0, True)):
        model, cfg, sim, torch = _sim(name, 1024, max_coord_velocity=clamp)
        rng = np.random.default_rng(7)
        acts = [torch.tensor(random_actions(model, rng, 1024), device="cuda:0") for _ in range(4)]
        stats = torch.zeros(8, device="cuda:0")
        for t in range(600):
            sim.step(acts[(t // 20) % 4], n_substeps=sub)
        sim.stats_flush(stats)
        assert bool(torch.isfinite(sim.get_state()).all())
        assert stats[0].item() == 600 * 1024
        assert (stats[2].item() > 0) == expect_resets, (name, clamp, stats.tolist())
        if clamp > 0:
            assert float(sim.get_state()[:, 7:13].abs().max()) <= clamp + 1e-3
        sim.close()


@pytest.mark.gpu
def test_env_groups_on_streams_are_bit_identical_to_one_handle():
    """SimGroups (G handles on G streams, no join between steps) must produce, env for env, exactly what one
    handle over all envs produces: envs are independent, only the schedule differs."""
    import torch
    from humanoid_gym_amd._capi import HgSim
    from humanoid_gym_amd.groups import SimGroups
    from humanoid_gym_amd.robots import load_robot
    model, cfg = load_robot("nao")
    n_envs = 50                                   # uneven split: 12 / 13 / 12 / 13
    one = HgSim(model, cfg, n_envs, "cuda:0")
    grp = SimGroups(model, cfg, n_envs, groups=4, device="cuda:0")
    assert grp.bounds[0] == 0 and grp.bounds[-1] == n_envs and len(grp.sims) == 4
    rng = np.random.default_rng(31)
    idx = np.asarray(model.act_joint)
    for t in range(40):
        act = torch.tensor(rng.uniform(model.lower[idx], model.upper[idx], (n_envs, model.n_act)),
                           dtype=torch.float32, device="cuda:0")
        one.step(act)
        grp.step(act)
        if t % 8 == 7:
            grp.wait()
            assert torch.equal(grp.obs_q, one.obs_q) and torch.equal(grp.obs_qd, one.obs_qd)
            assert torch.equal(grp.collision, one.collision) and torch.equal(grp.done, one.done)
    assert torch.equal(grp.get_state(), one.get_state())
    mask = torch.zeros(n_envs, dtype=torch.uint8, device="cuda:0")
    mask[5] = 1; mask[30] = 1
    one.reset(mask); grp.reset(mask)
    assert torch.equal(grp.get_state(), one.get_state())
    stats_g = torch.zeros(4, device="cuda:0"); stats_o = torch.zeros(4, device="cuda:0")
    grp.stats_flush(stats_g); one.stats_flush(stats_o)
    grp.wait(); torch.cuda.synchronize()
    assert torch.equal(stats_g, stats_o) and float(stats_o[0]) == 40 * n_envs
    grp.close(); one.close()


@pytest.mark.gpu
def test_wave_priority_flip_does_not_change_results():
    """hg_step flips the waves' arbiter priorities only for launches that fit the GPU at once.  A
    4096-env launch (flip on) and the first 4096 envs of a 4608-env launch (two rounds: flip off) must agree
    bit for bit: scheduling is not arithmetic."""
    import torch
    from humanoid_gym_amd._capi import HgSim
    from humanoid_gym_amd.robots import load_robot
    model, cfg = load_robot("nao")
    a, b = HgSim(model, cfg, 4096, "cuda:0"), HgSim(model, cfg, 4608, "cuda:0")
    rng = np.random.default_rng(77)
    idx = np.asarray(model.act_joint)
    for t in range(12):
        act = torch.tensor(rng.uniform(model.lower[idx], model.upper[idx], (4608, model.n_act)),
                           dtype=torch.float32, device="cuda:0")
        a.step(act[:4096].contiguous())
        b.step(act)
    torch.cuda.synchronize()
    assert torch.equal(a.get_state(), b.get_state()[:4096])
    assert torch.equal(a.obs_q, b.obs_q[:4096]) and torch.equal(a.collision, b.collision[:4096])
    a.close(); b.close()


@pytest.mark.gpu
def test_step_is_capturable_in_a_hip_graph():
    """hg_step allocates nothing and never synchronises the host, so a run of steps can be
    captured in a HIP graph and replayed: launch-bound loops (few envs) then cost one graph launch.  The replay
    must reproduce the eager run bit for bit."""
    import torch
    from humanoid_gym_amd._capi import HgSim
    from humanoid_gym_amd.robots import load_robot
    model, cfg = load_robot("nao")
    n_envs, K = 64, 8
    eager, graphed = HgSim(model, cfg, n_envs, "cuda:0"), HgSim(model, cfg, n_envs, "cuda:0")
    rng = np.random.default_rng(9)
    idx = np.asarray(model.act_joint)
    acts = torch.tensor(rng.uniform(model.lower[idx], model.upper[idx], (K, n_envs, model.n_act)),
                        dtype=torch.float32, device="cuda:0")
    static = acts.clone()
    graphed.step(static[0]); graphed.reset()          # warm up outside the capture (lazy module load)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(K):
            graphed.step(static[k])
    for rep in range(3):                              # 3 x K steps, same action rows each time
        g.replay()
        for k in range(K):
            eager.step(acts[k])
    torch.cuda.synchronize()
    assert torch.equal(eager.get_state(), graphed.get_state())
    assert torch.equal(eager.obs_q, graphed.obs_q) and torch.equal(eager.done, graphed.done)
    eager.close(); graphed.close()


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_gloo_rehearsal():
    """The N > 1 path end to end with the real simulator: `bench.py --gpus 2` starts two ranks itself; on a
    one-GPU box both bind cuda:0 (HG_BENCH_SINGLE_DEVICE) and the statistics all-reduce goes through gloo,
    which EpisodeStats must stage through the host (device tensors: the branch that once hung).
    RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HG_BENCH_SINGLE_DEVICE="1", HG_BENCH_SMALL="0", HG_STATS_EVERY="4", OMP_NUM_THREADS="4", HG_BENCH_CPU_SECONDS="2")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "12",
                          "--warmup", "4", "--envs-per-gpu", "256", "--stream-groups", "0"],
                         env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["envs_per_gpu"] == 256 and rec["config"]["state_finite"]
    assert rec["global_env_steps"] == 2 * 256 * (12 + 4)            # both ranks' counters arrived through the all-reduce
    assert rec["value"] > 0 and rec["per_rank_ms_per_step"]["max"] >= rec["per_rank_ms_per_step"]["min"] > 0
    # the N > 1 line is complete: backend and world size as torch.distributed saw them, the CPU baseline (rank 0), the roofline
    assert rec["backend"] == "gloo" and rec["world_size_seen"] == 2 and rec["config"]["n_dof"] == 42
    assert rec["cpu_baseline"]["value"] > 0 and rec["roofline"]["achieved"] > 0


def test_example_drivers_run(capsys):
    """examples/random_control.py and examples/h5_control.py (the reference's examples/*.py on this library): single env
    with the reference's return types, and batched."""
    import importlib.util
    mods = {}
    for name in ("random_control", "h5_control"):
        spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(HERE), "examples", name + ".py"))
        mods[name] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mods[name])
    obs = mods["random_control"].main(["dancer", "--steps", "20"])
    assert isinstance(obs, list) and len(obs) == 18                                  # dancer_env.py:52
    obs = mods["random_control"].main(["pepper", "--envs", "64", "--steps", "20"])
    assert tuple(obs["anglesPosition"].shape) == (64, 12)                           # pepper_env.py:64-68, batched
    obs = mods["h5_control"].main(["nao"])
    assert isinstance(obs, list) and len(obs) == 22 and np.isfinite(obs).all()       # nao_env.py:73
    obs = mods["h5_control"].main(["pepper", "--envs", "8"])
    assert np.isfinite(obs["anglesPosition"].cpu().numpy()).all()
    assert "replayed 240 frames" in capsys.readouterr().out
