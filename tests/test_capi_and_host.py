"""The C-ABI library loads on a CPU-only box and exports every symbol include/hg_sim.h
declares; host-side logic (workloads, stats sharding, spaces).  No GPU compute here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from humanoid_gym_amd import _capi, workloads
from humanoid_gym_amd.model import HgConfig, HgModel
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd.stats import EpisodeStats, shard_range
from humanoid_gym_amd import spaces

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "hg_sim.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(hg_[a-z_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    L = _capi.load_library()
    names = _declared_symbols()
    assert set(names) == set(_capi.SYMBOLS)
    for n in names:
        assert getattr(L, n) is not None
    from humanoid_gym_amd._capi import HG_ABI_VERSION
    assert L.hg_abi_version() == HG_ABI_VERSION == 8


def test_struct_sizes_match_header():
    """ctypes mirrors vs the C compiler's view of include/hg_model.h."""
    src = '#include <stdio.h>\n#include "hg_model.h"\nint main(){printf("%zu %zu\\n", sizeof(hg_model), sizeof(hg_config));return 0;}'
    exe = os.path.join("/tmp", "hg_sizes")
    subprocess.run(["gcc", "-x", "c", "-I", os.path.join(ROOT, "include"), "-o", exe, "-"], input=src.encode(), check=True)
    a, b = map(int, subprocess.check_output([exe]).split())
    assert a == C.sizeof(HgModel) and b == C.sizeof(HgConfig)


def test_error_paths_without_gpu():
    import torch
    L = _capi.load_library()
    h = C.c_void_p()
    assert L.hg_create(None, None, 1, 0, C.byref(h)) == -1            # HG_EINVAL
    assert b"null" in L.hg_last_error()
    m, cfg = load_robot("dancer")
    cm, cc = m.to_c(), cfg.to_c()
    assert L.hg_create(C.byref(cm), C.byref(cc), 0, 0, C.byref(h)) == -1
    cc.max_contacts = 99
    assert L.hg_create(C.byref(cm), C.byref(cc), 4, 0, C.byref(h)) == -1
    if not torch.cuda.is_available():
        cc = cfg.to_c()
        rc = L.hg_create(C.byref(cm), C.byref(cc), 4, 0, C.byref(h))
        assert rc == -2 and h.value is None                           # HG_ENODEVICE: no CPU fallback
        with pytest.raises(_capi.HgError):
            _capi.HgSim(m, cfg, 4)
    assert L.hg_step(None, None, None, 0, 1, None, None, None, None, None, None, None) == -1
    assert L.hg_destroy(None) == 0


def test_unsupported_trees_are_reported_on_the_host(tmp_path):
    """hg_create flattens the model and picks a kernel variant before it touches a device: a chain deeper
    than 12 links and three nested forks are HG_EUNSUPPORTED (-4), a child listed before its parent is
    HG_EINVAL (-1); supported trees (three forks in a row: the slots are recycled) get as far as the
    device probe."""
    import torch
    from _random_urdf import write_random_tree
    from humanoid_gym_amd.urdf import load_urdf
    from humanoid_gym_amd.model import SimConfig
    L = _capi.load_library()
    h = C.c_void_p()
    cc = SimConfig().to_c()

    def create(model):
        cm = model.to_c()
        return L.hg_create(C.byref(cm), C.byref(cc), 2, 0, C.byref(h))

    deep, _ = write_random_tree(str(tmp_path / "deep.urdf"), n_links=14, max_depth=14, n_chains=1, seed=1)
    assert create(load_urdf(deep, name="deep")) == -4 and b"deeper" in L.hg_last_error()
    ok, _ = write_random_tree(str(tmp_path / "ok.urdf"), n_links=27, max_depth=7, n_chains=3,
                              branch_at=((0, 2), (1, 2), (2, 2)), seed=5)
    m = load_urdf(ok, name="ok")
    nested = load_urdf(ok, name="nested")
    # links 0, 1 and 2 each have a second child that comes later in the order: three forks nested in one another
    nested.parent = np.array([-1, 0, 1, 2, 2, 1, 0] + [-1] * (nested.n_links - 7), np.int32)
    rc = create(nested)
    assert rc == -4 and b"nested" in L.hg_last_error(), (rc, L.hg_last_error())
    rc = create(m)
    if torch.cuda.is_available():
        assert rc == 0 and L.hg_destroy(h) == 0
    else:
        assert rc == -2
    cm = m.to_c()
    cm.parent[3] = 9                    # behind the back of FlatModel.check(): the library checks too
    assert L.hg_create(C.byref(cm), C.byref(cc), 2, 0, C.byref(h)) == -1 and b"parents-first" in L.hg_last_error()


def test_replay_workload_layout():
    m, _ = load_robot("nao")
    l, r = workloads.synthetic_h5_trajectory(m, T=240, seed=1234)
    assert l.shape == (240, 5) and r.shape == (240, 5) and l.dtype == np.float32
    rows = workloads.pack_rows(l, r)                                   # h5_control_nao.py:22
    assert rows.shape == (240, 12) and np.all(rows[:, 5] == 1) and np.all(rows[:, 11] == 1)
    np.testing.assert_array_equal(rows[:, :5], l); np.testing.assert_array_equal(rows[:, 6:11], r)
    names = workloads.nao_replay_joints()
    assert len(names) == 12 and names[5] == "LHand" and names[11] == "RHand"
    idx = [m.dof_of(n) for n in workloads.L_ARM]
    assert np.all(l >= m.lower[idx] - 1e-6) and np.all(l <= m.upper[idx] + 1e-6)
    path = "/tmp/hg_traj_test.npz"
    workloads.save_trajectory(path, l, r)
    l2, r2 = workloads.load_trajectory(path)
    np.testing.assert_array_equal(l, l2); np.testing.assert_array_equal(r, r2)
    a = workloads.random_actions(m, np.random.default_rng(0), 7)
    assert a.shape == (7, 22) and np.all(a >= m.lower[m.act_joint] - 1e-6) and np.all(a <= m.upper[m.act_joint] + 1e-6)


def test_shard_range_covers_everything_once():
    for total, world in ((4096, 8), (10, 3), (5, 8), (0, 2)):
        seen = []
        for r in range(world):
            lo, hi = shard_range(total, r, world)
            assert 0 <= lo <= hi <= total
            seen += list(range(lo, hi))
        assert seen == list(range(total))


def test_stats_single_process_cpu():
    import torch
    st = EpisodeStats(None, reduce_every=2, device="cpu")
    for t in range(4):
        st.accumulate(reward=torch.ones(5), done=torch.tensor([1, 0, 0, 0, 0]), collision=torch.tensor([1, 1, 0, 0, 0]))
        if (t + 1) % 2 == 0:
            st.reduce_async()
    assert st.flush().tolist()[:4] == [20.0, 20.0, 4.0, 8.0]


def test_stats_allreduce_world2_gloo():
    """The N > 1 path: two ranks, disjoint env slices, gloo all-reduce of the 8-float vector."""
    script = os.path.join(ROOT, "tests", "_gloo_stats_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", script],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GLOO_STATS_OK" in out.stdout


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` with no RANK in the environment must start two ranks itself (BASELINE.json:
    "at 1/2/4/8 MI355X") and print ONE line with n_gpus = 2.  --dry-sim swaps HgSim for a stand-in so that the
    launcher, the env sharding and the statistics all-reduce run on CPU ranks over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-sim", "--steps", "3",
                          "--warmup", "1", "--envs-per-gpu", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["backend"] == "gloo"
    assert rec["world_size_seen"] == 2                                # what torch.distributed itself reports
    assert rec["global_env_steps"] == 2 * 64 * (3 + 1)              # both ranks' envs went through the all-reduce
    assert rec["per_rank_ms_per_step"]["min"] <= rec["per_rank_ms_per_step"]["max"]
    assert "DRY RUN" in rec["metric"] and rec["value"] == 0.0        # never mistaken for a measurement


def test_box_space_fallback():
    b = spaces.Box(np.array([-1.0, 0.0], dtype=np.float32), np.array([1.0, 2.0], dtype=np.float32))
    for _ in range(10):
        assert b.contains(b.sample())


def test_reference_robot_kernel_variants_use_no_scratch():
    """build() records hipcc's resource remarks per instantiation of the step kernel (csrc/kernel_resources.json).
    The variants that the reference's four robots run on keep everything in registers and LDS:
    no scratch, and the occupancy the LDS plan is sized for."""
    import json
    from humanoid_gym_amd import build as hb
    from humanoid_gym_amd.robots import load_robot
    if not os.path.exists(hb.RESOURCES) or hb.needs_build():
        pytest.skip("kernel_resources.json is written by humanoid_gym_amd.build.build(); library not built from these sources")
    res = json.load(open(hb.RESOURCES))
    expect = {"dancer": ("<18,12,6,0,4>", 4), "nao": ("<26,8,6,0,4>", 4), "pepper": ("<20,8,9,2,4>", 4),
              "romeo": ("<37,8,12,2,3>", 3), "nao_hands": ("<42,8,12,2,3>", 3)}
    for robot, (variant, waves) in expect.items():
        model, cfg = load_robot(robot)
        nmx, maxc = (int(x) for x in variant.strip("<>").split(",")[:2])
        assert model.n_links <= nmx and cfg.max_contacts <= maxc, robot          # the shape fits the variant named here
        r = res[variant]
        assert r["scratch_bytes_per_lane"] == 0, (robot, variant, r)
        assert r["occupancy_waves_per_simd"] == waves, (robot, variant, r)


def test_parity_bookkeeping_helpers():
    """The helpers the GPU parity tests lean on, exercised on the CPU: FloatParity accepts a contact sample beyond the
    tolerance only when the oracle's own spread under float32-sized row noise explains it (bounded, counted), and
    nothing else; IntParity accepts a different contact set only with margin slack."""
    from test_gpu_parity import FloatParity, IntParity, SLACK_TOL, EXPLAIN_FACTOR
    wide = dict(q=1e-3, base=0.0, qd=0.2, tau_rel=0.1)           # an oracle answer that jumps under row noise
    steady = dict(q=4e-7, base=1e-7, qd=1e-4, tau_rel=1e-5)      # an ordinary sample
    fp = FloatParity(1e-4, 0.03)
    for _ in range(300):
        fp.add(4, 2e-6, 1e-3)
    fp.add(12, 3e-3, 0.5, spread=wide)       # one degenerate contact sample, explained
    fp.report("unit")                        # 1 <= max(1, ceil(0.003 * 301)) = 1
    fp.add(7, 2e-3, spread=wide)
    with pytest.raises(AssertionError):
        fp.report("unit")                    # two explained outliers among 302 contact samples: one too many
    fq = FloatParity(1e-4, 0.03)             # beyond a tolerance by LESS than the oracle's own spread: counted apart, bounded at 2 %
    for _ in range(100):
        fq.add(4, 2e-6, 1e-3)
    fq.add(5, 5e-4, 0.1, spread=wide); fq.add(5, 9e-4, 0.19, spread=wide)
    assert fq.noisy == 2 and fq.outliers == []
    fq.report("unit")
    fq.add(5, 6e-4, 0.1, spread=wide); fq.add(5, 6e-4, 0.1, spread=wide)
    with pytest.raises(AssertionError):
        fq.report("unit")                    # 4 > max(2, 2 % of 104)
    ff = FloatParity(1e-4, explain_contact_free=True)      # the fuzzers' mode: a contact-free sample may be explained too
    ff.add(0, 2e-4, spread=dict(q=1e-4, base=0.0, qd=0.0, tau_rel=0.0))
    assert len(ff.outliers) == 1
    with pytest.raises(AssertionError):
        FloatParity(1e-4).add(3, 5e-4, spread=steady)      # beyond the tolerance and NOT explained: this is what a kernel bug looks like
    with pytest.raises(AssertionError):
        FloatParity(1e-4).add(3, 1.01 * EXPLAIN_FACTOR * 1e-3, spread=wide)      # further off than the spread explains
    with pytest.raises(AssertionError):
        FloatParity(1e-4).add(3, 5e-4)       # an outlier with nothing to explain it by
    with pytest.raises(AssertionError):
        FloatParity(1e-4).add(0, 2e-4, spread=wide)       # a contact-free sample may never be an outlier
    with pytest.raises(AssertionError):
        FloatParity(1e-4).add(3, 0.5, spread=wide)        # beyond the bound
    with pytest.raises(AssertionError):
        FloatParity(1e-4).add(3, float("nan"))
    # the fuzzers' mode: within the oracle's OWN spread the spread is the bound; a marginal explanation stays under `bound`
    chaotic = dict(q=0.135, base=0.03, qd=32.0, tau_rel=2.0)
    fz = FloatParity(2e-4, bound_within_own_spread=False)
    fz.add(3, 0.058, spread=chaotic)
    assert fz.noisy == 1 and fz.outliers == []
    with pytest.raises(AssertionError):
        fz.add(3, 0.058, spread=dict(q=0.02, base=0.0, qd=5.0, tau_rel=0.1))      # 2.9x the spread: marginal, and beyond the bound
    with pytest.raises(AssertionError):
        FloatParity(2e-4).add(3, 0.058, spread=chaotic)                           # the tests keep the bound for every sample

    class _Cpu:
        n_envs = 2
        def __init__(self, slack): self.slack = slack
        def contact_slack(self, e): return (self.slack[e], 1e30)
    o = dict(n_contacts=np.array([1, 1]), pairs=np.zeros((2, 12, 2), np.int32), link_mask=np.array([2, 2], np.uint64),
             collision=np.array([1, 1], np.uint8))
    counts, pairs, mask = np.array([1, 2]), np.zeros((2, 12, 2), np.int32), np.array([2, 2], np.uint64)
    ip = IntParity(); ip.pre(_Cpu([1.0, SLACK_TOL / 2]))
    assert ip.check(0, o, counts, pairs, mask) and not ip.check(1, o, counts, pairs, mask)      # env 1 differs, explained by slack
    ip = IntParity(); ip.pre(_Cpu([1.0, 1e-3]))
    with pytest.raises(AssertionError):
        ip.check(1, o, counts, pairs, mask)                                                     # differs, nothing near the margin


def test_hg_create_rejects_numbers_the_step_cannot_use():
    """A NaN mass or a zero inertia does not fault on the device: every env of the handle turns non-finite and the guard
    resets it on every step, silently.  hg_create checks the numbers on the host (before it looks for a device)."""
    import math
    L = _capi.load_library()
    h = C.c_void_p()
    for name in ("dancer", "nao", "pepper", "romeo", "nao_hands", "romeo_hands"):      # the shipped models pass the checks
        m, cfg = load_robot(name)
        cm, cc = m.to_c(), cfg.to_c()
        rc = L.hg_create(C.byref(cm), C.byref(cc), 2, 0, C.byref(h))
        assert rc in (0, -2), (name, rc, L.hg_last_error())
        if rc == 0:
            L.hg_destroy(h)
    m, cfg = load_robot("nao")

    def rejected(edit_model=None, edit_cfg=None):
        cm, cc = m.to_c(), cfg.to_c()
        if edit_model:
            edit_model(cm)
        if edit_cfg:
            edit_cfg(cc)
        rc = L.hg_create(C.byref(cm), C.byref(cc), 2, 0, C.byref(h))
        assert rc == -1 and h.value is None, (rc, L.hg_last_error())
        return L.hg_last_error()

    def put(field, i, v, k=None):
        def edit(cm):
            if i is None:
                setattr(cm, field, v)
            elif k is None:
                getattr(cm, field)[i] = v
            else:
                getattr(cm, field)[i][k] = v
        return edit

    assert b"link mass" in rejected(put("mass", 3, float("nan")))
    assert b"link mass" in rejected(put("mass", 0, 0.0))
    assert b"(index 5)" in rejected(put("inertia", 5, -1e-4, 1))
    assert b"base mass" in rejected(put("base_mass", None, math.inf))
    assert b"unit vector" in rejected(lambda cm: [cm.axis[2].__setitem__(k, 0.0) for k in range(3)])
    assert b"unit quaternion" in rejected(put("base_quat0", 3, 2.0))
    assert b"lower <= upper" in rejected(lambda cm: (cm.lower.__setitem__(1, 1.0), cm.upper.__setitem__(1, -1.0)))
    assert b"bounds must be >= 0" in rejected(put("max_force", 4, -1.0))
    assert b"collision primitive" in rejected(put("sphere_radius", 0, float("nan")))
    assert b"frame constants" in rejected(put("d", 7, float("inf"), 2))
    assert b"config" in rejected(edit_cfg=lambda cc: setattr(cc, "friction", float("nan")))
    assert b"config" in rejected(edit_cfg=lambda cc: cc.gravity.__setitem__(2, float("inf")))
    assert b"config" in rejected(edit_cfg=lambda cc: setattr(cc, "kp", -0.1))
    if len(m.pair_a):
        assert b"within one body" in rejected(lambda cm: cm.pair_b.__setitem__(0, cm.pair_a[0]))
    # a ground primitive is a point or a sphere: segments take part in link-link pairs only
    ground = next(k for k in range(len(m.sphere_link)) if m.sphere_flags[k] & 1)
    assert b"point or a sphere" in rejected(put("sphere_half", ground, 0.01, 2))
    # Bullet's residual early-out lives in the oracle only (measured never to trigger: DESIGN 1); the library says so
    # instead of ignoring the field
    cm, cc = m.to_c(), cfg.to_c()
    cc.residual_threshold = 1e-7
    assert L.hg_create(C.byref(cm), C.byref(cc), 2, 0, C.byref(h)) == -4 and b"residual_threshold" in L.hg_last_error()
