#!/usr/bin/env python3
"""bench.py -- NAO env-steps/s at 4096 envs per GPU (BASELINE.json `metric`).

The robot of `value` is the NAO stand-in WITH the 16 finger / thumb joints of its hands (42 dofs, 66 constraint rows),
the size at which the reference simulates the robot: qibullet's URDF has them as dofs and the reference's envs only keep
them out of the CONTROLLED joint lists (romeo_env.py:24-26).  The fingers follow LHand / RHand through their <mimic>
columns, as qibullet's setAngles drives them.  The smaller 26-dof stand-in without finger joints (the headline of
rounds 1-2) is measured beside it as `nao_without_finger_joints`.

Workload (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): the NAO
joint-trajectory replay of /root/reference/examples/h5_control_nao.py with the
`inference.h5` file (not in the reference tree) replaced by a synthetic
trajectory of the same layout: l_joint_angle[T,5], r_joint_angle[T,5], action
row = l ++ [1] ++ r ++ [1] (h5_control_nao.py:22), applied to the 10 arm joints
+ LHand/RHand through NaoEnv.step's `joints=` argument (nao_env.py:48,70).
Every env replays the same trajectory with a phase offset env_id mod T.
PD position control + ground contact on, 1 physics substep (1/240 s) per step.

One "step" = one hg_step launch over all envs of this rank (targets -> physics
-> observations, with the per-env episode-statistics accumulation fused into the
kernel's output stage); actions are resident in HBM before the timed region.  N > 1: one process per GPU,
envs sharded with no data-path collective; the only collective is hg_stats_flush + the RCCL all-reduce of
the 8-float episode-statistics vector every 64 steps on a side stream.  `python bench.py --gpus N` with no
RANK in the environment starts the N ranks itself (a child `python -m torch.distributed.run`, the parent
never touches a GPU) and relays rank 0's line; under an outer torchrun it is one of the ranks.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_TRAJ = 240
ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector


def synthetic_replay(model, seed=1234):
    """[T, 12] float32 action rows in the h5_control_nao layout."""
    from humanoid_gym_amd.workloads import nao_replay_joints, synthetic_h5_trajectory
    names = nao_replay_joints()
    l, r = synthetic_h5_trajectory(model, T=T_TRAJ, seed=seed)
    rows = np.concatenate([l, np.ones((T_TRAJ, 1), np.float32), r, np.ones((T_TRAJ, 1), np.float32)], axis=1)
    sel = np.array([model.dof_of(n) for n in names], np.int32)
    return rows.astype(np.float32), sel


def with_mimic_columns(model, rows, sel, names):
    """qibullet's setAngles drives the finger joints that <mimic> LHand / RHand together with them (the env layer
    does the same, softbank_envs.py): one more action column per such joint, target = multiplier * hand + offset."""
    cols, sel = [rows], list(sel)
    for child, (parent, mult, off) in (model.meta.get("mimic") or {}).items():
        if parent in names and child in model.dof_names:
            sel.append(model.dof_of(child))
            cols.append((mult * rows[:, names.index(parent)] + off)[:, None])
    return np.concatenate(cols, axis=1).astype(np.float32), np.array(sel, np.int32)


def algorithmic_bytes_per_env_step(model, n_sel):
    """actions + obs + reward + flags + state read and written once."""
    n = model.n_links
    return 4 * (n_sel + model.n_act) + 4 + 2 + 2 * 4 * (13 + 2 * n)


def algorithmic_flops_per_env_step(model, cfg, n_contacts=8, n_substeps=1):
    """Flops of the algorithm as the oracle states it (Bullet's velocity-space form), see DESIGN.md:
    ABA ~300/body, one M^-1 J^T response ~150/body per row, per PGS row a dot + an axpy of length 6+n."""
    nb, ndof = model.n_links + 1, 6 + model.n_links
    rows = model.n_links + 3 * n_contacts
    return n_substeps * (300 * nb + rows * 150 * nb + cfg.solver_iters * rows * 4 * ndof)


def cpu_baseline(model, cfg, rows, sel, target_seconds=float(os.environ.get("HG_BENCH_CPU_SECONDS", "12"))):
    """Oracle (oracle/hg_oracle.c, FP64, OpenMP over envs) on this host, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import OracleSim
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    n_envs = 64 * threads
    sim = OracleSim(model, cfg, n_envs)
    phase = np.arange(n_envs) % T_TRAJ

    def run(steps, t0):
        for t in range(steps):
            sim.step(rows[(t0 + t + phase) % T_TRAJ], joint_sel=sel)
    run(4, 0)                                   # thread pool + cache warm-up
    c0 = time.perf_counter(); run(4, 4); per = (time.perf_counter() - c0) / 4
    steps = int(max(8, min(400, target_seconds / max(per, 1e-6))))
    c0 = time.perf_counter(); run(steps, 8); dt = time.perf_counter() - c0
    out = {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
           "sample": f"oracle FP64 OpenMP, {n_envs} envs x {steps} steps of the same NAO replay workload"}
    # single thread beside it = BASELINE.json configs[0] (SURVEY 8d "Config 1"): nao-v0, ONE env, random_control
    # (random_control_nao.py:9: action_space.sample() each step = U(lower_j, upper_j), default_rng(0)), 1000 steps;
    # the same robot model as `value` (configs[0] names nao-v0: here the model with the finger joints);
    # the oracle takes its serial path below four envs
    one = OracleSim(model, cfg, 1)
    idx = np.asarray(model.act_joint)
    rng = np.random.default_rng(0)
    acts = rng.uniform(model.lower[idx], model.upper[idx], (1000, 1, len(idx)))
    c0 = time.perf_counter()
    for t in range(1000):
        one.step(acts[t])
    # flops of the same workload as the oracle's instrumented build counts them, beside the formula that
    # `valu` is priced with; one env, the second pass over the trajectory (standing on both feet)
    from _oracle import counted_flops_per_env_step
    cnt = counted_flops_per_env_step(model, cfg, np.concatenate([rows, rows]), sel, warmup=T_TRAJ)
    out["counted_flops_per_env_step"] = {"value": cnt["flops_per_env_step"], "mean_contacts": cnt["mean_contacts"],
                                         "formula": algorithmic_flops_per_env_step(model, cfg, n_contacts=round(cnt["mean_contacts"])),
                                         "note": "oracle/libhg_oracle_flops.so (-DHGO_COUNT_FLOPS): multiplies + adds the FP64 "
                                                 "restatement executes per env step; `valu` uses the formula"}
    out["single_thread"] = {"value": 1000 / (time.perf_counter() - c0), "unit": "env-steps/s",
                            "sample": "BASELINE configs[0]: nao, 1 env x 1000 random_control steps, oracle FP64, "
                                      "includes the ctypes call per step"}
    return out


def launch_ranks(argv, n):
    """--gpus N without a launcher around us: start N ranks as a child torchrun and relay its output.
    This process has not initialised HIP (no exec after GPU init: the ranks are CHILD processes)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


class DrySim:
    """--dry-sim: stands in for HgSim so that the launcher, the sharding and the statistics all-reduce can be
    rehearsed on CPU ranks (gloo); it simulates nothing and its line says so."""

    def __init__(self, model, n_envs, device):
        import torch
        self.device, self.n_envs, self.n_act = torch.device(device), n_envs, model.n_act
        self.reward = torch.zeros(n_envs); self.done = torch.zeros(n_envs, dtype=torch.uint8)
        self.collision = torch.ones(n_envs, dtype=torch.uint8)
        self._steps = 0

    def step(self, actions=None, joint_sel=None, n_substeps=1):
        self._steps += 1

    def stats_flush(self, stats):
        stats[0] += self._steps * self.n_envs; stats[3] += self._steps * self.n_envs
        self._steps = 0
        return stats


def _dbg(msg):
    if os.environ.get("HG_BENCH_DEBUG"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner to
    stdout when the box exports NCCL_DEBUG=VERSION), so from here on file descriptor 1 of this process is stderr, and the
    file object returned -- the real stdout -- is used for the JSON line only."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--robot", default="nao_hands", help="nao_hands (default: the NAO at the size the reference simulates), nao (26 dofs, no finger joints), pepper, romeo, romeo_hands, dancer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cfg", action="append", default=[], metavar="KEY=VALUE",
                    help="override a SimConfig field (diagnostics only; the reported metric is the default config)")
    ap.add_argument("--stream-groups", type=int, default=2,
                    help="env groups of the extra `pipelined` measurement (0 = skip it); `value` is always one launch per step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only to rehearse")
    ap.add_argument("--dry-sim", action="store_true",
                    help="CPU rehearsal of the N > 1 path (launcher, sharding, statistics all-reduce) without a GPU: "
                         "no simulation, the line is labelled and is not a measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(sys.argv[1:], args.gpus))
    json_out = claim_stdout()
    if os.environ.get("HG_BENCH_DEBUG"):     # where does a stuck rank sit?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["HG_BENCH_DEBUG"]), exit=True)

    import ast
    import torch
    import torch.distributed as dist
    from humanoid_gym_amd.robots import load_robot
    from humanoid_gym_amd.stats import EpisodeStats

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("HG_BENCH_SINGLE_DEVICE"):     # rehearsal of the N > 1 path on a one-GPU box
        local_rank = 0
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)   # torchrun, any N
    dry = args.dry_sim
    if dry:
        args.backend = "gloo"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not dry:
            torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    if not dry:
        torch.cuda.set_device(dev)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    _dbg("process group ready")

    model, cfg = load_robot(args.robot)
    for kv in args.cfg:
        k, v = kv.split("=", 1)
        setattr(cfg, k, type(getattr(cfg, k))(ast.literal_eval(v)))
    n_envs = args.envs_per_gpu
    is_nao = args.robot in ("nao", "nao_hands")
    if is_nao and not any(kv.startswith("store_torques=") for kv in args.cfg):
        cfg.store_torques = False      # as NaoEnv (envs/softbank_envs.py): nothing on this path reads getJointState()[3]
    if is_nao and not any(kv.startswith("store_contacts=") for kv in args.cfg):
        cfg.store_contacts = False     # ... nor getContactPoints (the `collision` flag is written either way)
    if is_nao:
        rows, sel = synthetic_replay(model)
        if args.robot == "nao_hands":
            from humanoid_gym_amd.workloads import nao_replay_joints
            rows, sel = with_mimic_columns(model, rows, sel, nao_replay_joints())
    else:   # other robots: random targets inside the limits, all controlled joints
        rng = np.random.default_rng(2)
        sel = np.asarray(model.act_joint, np.int32)
        rows = rng.uniform(model.lower[sel], model.upper[sel], (T_TRAJ, len(sel))).astype(np.float32)
    if dry:
        sim = DrySim(model, n_envs, dev)
    else:
        from humanoid_gym_amd._capi import HgSim
        sim = HgSim(model, cfg, n_envs, dev)
        if is_nao:
            sim.obs_qd = None          # NaoEnv's observation is the joint positions only (nao_env.py:73-79)
    # actions for every (time, env) resident in HBM: [T, n_envs, n_sel]
    first_env = rank * n_envs
    phase = (torch.arange(n_envs, device=dev) + first_env) % T_TRAJ
    rows_d = torch.tensor(rows, device=dev)
    actions = torch.stack([rows_d[(t + phase) % T_TRAJ] for t in range(T_TRAJ)]).contiguous()
    # statistics all-reduce every 64 steps on a side stream; HG_STATS_EVERY / HG_STATS_SIDE=0 (main stream) for tools/stats_cadence.sh.
    # One RCCL rank, 4096 NAO envs: every 32 side / main stream 0.1164 / 0.1158 ms per step, every 64
    # 0.1159 / 0.1151, no process group 0.1139.  The side stream bounds the cost by the one step that the collective's
    # kernel displaces, whatever the latency of an 8-rank ring turns out to be.
    stats = EpisodeStats(sim, reduce_every=int(os.environ.get("HG_STATS_EVERY", "64")), side_stream=os.environ.get("HG_STATS_SIDE", "1") != "0")
    _dbg("sim and actions ready")

    def step(t):
        sim.step(actions[t % T_TRAJ], joint_sel=sel, n_substeps=1)
        stats.update(t)

    for t in range(args.warmup):
        step(t)
    stats.flush()          # one statistics all-reduce outside the timed region (RCCL sets up its channels lazily)
    sync()
    _dbg("warmup done")
    if use_dist:
        dist.barrier()
    sync()
    # HIP events on the stream the kernel is launched on, around the timed steps themselves: the step kernel's average
    # launch duration as the roofline uses it (includes the ~2 us between launches and the 64-byte statistics flushes)
    ev_a = ev_b = None
    if not dry:
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if ev_a is not None:
        ev_a.record()
    for t in range(args.warmup, args.warmup + args.steps):
        step(t)
    if ev_b is not None:
        ev_b.record()
    stats.flush()
    sync()
    own = time.perf_counter() - t0          # this rank's own work, before it waits for the others
    if use_dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    _dbg("timed region done")
    per_rank_ms = [own / args.steps * 1e3]
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        mine = torch.tensor([own / args.steps * 1e3], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank_ms = [float(g.item()) for g in gathered]
    global_env_steps = float(stats.total[0].item())      # all-reduced: every rank's envs x (warmup + steps)

    if dry:
        if rank == 0:
            print(json.dumps({"metric": "DRY RUN (no simulation): launcher / sharding / all-reduce rehearsal", "value": 0.0,
                              "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "none", "data": "none", "backend": args.backend,
                              "world_size_seen": (dist.get_world_size() if use_dist else 1),
                              "per_rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms)},
                              "global_env_steps": global_env_steps,
                              "config": {"workload": "dry", "envs_per_gpu": n_envs, "first_env_of_rank0": first_env}}),
                  file=json_out, flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    kernel_ms = ev_a.elapsed_time(ev_b) / args.steps          # over the timed region (both events have completed: synchronised above)
    # and back-to-back launches of the step kernel alone, for comparison with the rocprofv3 kernel statistics
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kreps = max(10, min(args.steps, 100))
    ev0.record()
    for t in range(kreps):
        sim.step(actions[t % T_TRAJ], joint_sel=sel, n_substeps=1)
    ev1.record()
    torch.cuda.synchronize()
    kernel_ms_b2b = ev0.elapsed_time(ev1) / kreps
    if cfg.store_contacts:
        _, counts, _ = sim.contact_pairs()
    else:       # the timed handle keeps no contact lists: one more step of a copy that does, from the same state
        import dataclasses
        chk = HgSim(model, dataclasses.replace(cfg, store_contacts=True), n_envs, dev)
        chk.set_state(sim.get_state())
        chk.step(actions[kreps % T_TRAJ], joint_sel=sel, n_substeps=1)
        _, counts, _ = chk.contact_pairs()
        chk.close()
    mean_contacts = float(counts.float().mean().item())

    # Beside the headline: the same envs as G handles on G streams, not joined between steps
    # (humanoid_gym_amd/groups.py) -- the replay has no observation -> action dependency, so steps of
    # different groups may overlap; same work per step, statistics flushed at the same cadence.
    pipelined = None
    if args.stream_groups > 1 and world == 1:     # extras are measured on a single-GPU run only
        try:          # an extra beside the headline: it must never take the bench line down with it
            from humanoid_gym_amd.groups import SimGroups
            grp = SimGroups(model, cfg, n_envs, args.stream_groups, dev)
            gstats = torch.zeros(4, dtype=torch.float32, device=dev)

            def gstep(t):
                grp.step(actions[t % T_TRAJ], joint_sel=sel, n_substeps=1, order_after_caller=False)
                if (t + 1) % 16 == 0:
                    grp.stats_flush(gstats)
            for t in range(args.warmup):
                gstep(t)
            grp.wait(); torch.cuda.synchronize()
            g0 = time.perf_counter()
            for t in range(args.warmup, args.warmup + args.steps):
                gstep(t)
            grp.stats_flush(gstats)
            grp.wait(); torch.cuda.synchronize()
            g_el = time.perf_counter() - g0
            pipelined = {"stream_groups": args.stream_groups, "value_this_rank": n_envs * args.steps / g_el,
                         "unit": "env-steps/s", "ms_per_step": g_el / args.steps * 1e3,
                         "state_finite": bool(torch.isfinite(grp.get_state()).all().item()),
                         "note": "same envs and steps as `value`, split into handles on their own streams and not "
                                 "joined between steps; `value` itself is one launch per step"}
            grp.close()
        except Exception as e:      # noqa: BLE001
            pipelined = {"error": f"{type(e).__name__}: {e}"}
    finite = bool(torch.isfinite(sim.get_state()).all().item())

    # Beside the headline as well: the same replay on the smaller NAO stand-in WITHOUT finger joints (26 dofs, 50
    # constraint rows), the headline robot of rounds 1-2.
    without_hands = None
    if args.robot == "nao_hands" and not args.cfg and world == 1 and os.environ.get("HG_BENCH_SMALL", "1") != "0":
        try:
            hmodel, hcfg = load_robot("nao")
            hcfg.store_torques, hcfg.store_contacts = cfg.store_torques, cfg.store_contacts
            hrows, hsel = synthetic_replay(hmodel)
            hsim = HgSim(hmodel, hcfg, n_envs, dev)
            hsim.obs_qd = None
            hrows_d = torch.tensor(hrows, device=dev)
            hact = torch.stack([hrows_d[(t + phase) % T_TRAJ] for t in range(T_TRAJ)]).contiguous()
            for t in range(args.warmup):
                hsim.step(hact[t % T_TRAJ], joint_sel=hsel, n_substeps=1)
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            for t in range(args.warmup, args.warmup + args.steps):
                hsim.step(hact[t % T_TRAJ], joint_sel=hsel, n_substeps=1)
            torch.cuda.synchronize()
            h_el = time.perf_counter() - h0
            without_hands = {"robot": "nao_like (stand-in URDF without finger joints)", "n_dof": hmodel.n_links,
                             "constraint_rows": hmodel.n_links + 3 * hcfg.max_contacts, "action_columns": int(len(hsel)),
                             "value_this_rank": n_envs * args.steps / h_el, "unit": "env-steps/s",
                             "ms_per_step": h_el / args.steps * 1e3,
                             "state_finite": bool(torch.isfinite(hsim.get_state()).all().item())}
            hsim.close()
        except Exception as e:      # noqa: BLE001
            without_hands = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        total_envs = n_envs * world
        value = total_envs * args.steps / elapsed
        bytes_step = algorithmic_bytes_per_env_step(model, len(sel))
        flops_step = algorithmic_flops_per_env_step(model, cfg, n_contacts=int(round(mean_contacts)))
        # HBM bytes per launch: PMC counters cannot be read from inside this process, so the figure is the one
        # of the committed rocprofv3 --pmc passes over this same kernel and configuration (named in traffic_source)
        traffic, traffic_source = None, None
        pmc_file = {"nao_hands": "round3_pmc_nao_hands4096.json", "nao": "round3_pmc_nao4096.json"}.get(args.robot)
        try:
            if pmc_file and n_envs == 4096 and not args.cfg:
                pm = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
                traffic = pm["per_launch"].get("hbm_bytes_corrected", pm.get("hbm_bytes_corrected"))
                traffic_source = f"profiles/{pmc_file} (rocprofv3 --pmc passes of this kernel and config, not this run)"
        except Exception:
            traffic, traffic_source = None, None
        ach_gbs = n_envs * bytes_step / (kernel_ms * 1e-3) / 1e9
        ach_tf = n_envs * flops_step / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "NAO env-steps/sec (4096 envs/GPU)" if is_nao else f"{args.robot} env-steps/sec",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "per_rank_ms_per_step": {"min": min(per_rank_ms), "max": max(per_rank_ms)},
            "global_env_steps": global_env_steps,
            "backend": (args.backend if use_dist else "none (single process)"),
            "world_size_seen": (dist.get_world_size() if use_dist else 1),
            "config": {"workload": "NAO h5 joint-trajectory replay (h5_control_nao layout, synthetic sinusoid "
                                   "trajectory), PD position control + ground contact" + (" + link-link contacts" if cfg.enable_self else "") + ", 1 substep of 1/240 s"
                       if is_nao else f"{args.robot} random targets",
                       "robot": model.name + ("_like (stand-in URDF" + (", with the finger joints of the hands" if args.robot.endswith("_hands") else "") + ")" if args.robot != "dancer" else ""),
                       "constraint_rows": model.n_links + 3 * cfg.max_contacts, "action_columns": int(len(sel)),
                       "envs_per_gpu": n_envs, "n_dof": model.n_links, "solver_iters": cfg.solver_iters, "motor_torques_stored": bool(cfg.store_torques), "contact_lists_stored": bool(cfg.store_contacts),
                       "mean_contacts": mean_contacts, "state_finite": finite, "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "hg_step_kernel", "kernel_ms": kernel_ms, "kernel_ms_back_to_back": kernel_ms_b2b,
                         "kernel_ms_note": "kernel_ms: HIP events around the timed steps on the launch stream / steps; "
                                           "back_to_back: the same kernel alone, 100 launches",
                         "algorithmic_bytes_per_env_step": bytes_step,
                         "note": "metric asks for the HBM fraction; the kernel is instruction-issue bound (FP32 VALU pipe, issue slots of the solver sweeps), see `valu` and DESIGN.md section 4"},
            "valu": {"achieved": ach_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / VALU_PEAK_TFLOPS,
                     "algorithmic_flops_per_env_step": flops_step},
        }
        if pipelined is not None:
            out["pipelined"] = pipelined
        if without_hands is not None:
            out["nao_without_finger_joints"] = without_hands
        if not args.no_cpu_baseline:         # rank 0 only, after the timed region and its closing barrier, for every N
            out["cpu_baseline"] = cpu_baseline(model, cfg, rows, sel)
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
