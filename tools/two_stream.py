"""Does splitting the envs of one GPU into G groups, each stepped on its own stream without joining the
streams between steps, recover the out-of-phase steady state?  (A single 4096-env launch is ONE resident
round in which all waves of a SIMD run the same phase at the same time.)   usage: two_stream.py [envs] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from humanoid_gym_amd.robots import load_robot
from humanoid_gym_amd._capi import HgSim

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
model, cfg = load_robot("nao")
rows, sel = bench.synthetic_replay(model)
rows_d = torch.tensor(rows, device=dev)
T = rows.shape[0]

for groups in (1, 2, 4):
    n = n_total // groups
    sims = [HgSim(model, cfg, n, dev) for _ in range(groups)]
    streams = [torch.cuda.Stream(dev) for _ in range(groups)]
    acts = []
    for g in range(groups):
        phase = (torch.arange(n, device=dev) + g * n) % T
        acts.append(torch.stack([rows_d[(t + phase) % T] for t in range(T)]).contiguous())
    torch.cuda.synchronize()

    def run(k, t0):
        for t in range(k):
            for g in range(groups):
                with torch.cuda.stream(streams[g]):
                    sims[g].step(acts[g][(t0 + t) % T], joint_sel=sel)
    run(40, 0)
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    run(steps, 40)
    torch.cuda.synchronize()
    dt = time.perf_counter() - c0
    print(f"{groups} group(s) x {n} envs: {n_total * steps / dt / 1e6:.2f} M env-steps/s  ({dt / steps * 1e6:.1f} us per step of all groups)", flush=True)
    for s in sims:
        s.close()

# the same through SimGroups (shared output tensors, prebuilt C arguments): host issue time and total time
from humanoid_gym_amd.groups import SimGroups
phase = torch.arange(n_total, device=dev) % T
actions = torch.stack([rows_d[(t + phase) % T] for t in range(T)]).contiguous()
for groups in (2, 4, 8):
    for flush in (0, 16):
        grp = SimGroups(model, cfg, n_total, groups, dev)
        gstats = torch.zeros(4, device=dev)
        def grun(k, t0):
            for t in range(t0, t0 + k):
                grp.step(actions[t % T], joint_sel=sel, order_after_caller=False)
                if flush and (t + 1) % flush == 0:
                    grp.stats_flush(gstats)
        grun(40, 0)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        grun(steps, 40)
        c1 = time.perf_counter()
        torch.cuda.synchronize()
        c2 = time.perf_counter()
        print(f"SimGroups {groups} groups, stats flush every {flush}: {n_total * steps / (c2 - c0) / 1e6:.2f} M env-steps/s, "
              f"host issue {(c1 - c0) / steps * 1e6:.1f} us/step, total {(c2 - c0) / steps * 1e6:.1f} us/step", flush=True)
        grp.close()
