"""Wall-clock cost of the Python env layer around hg_step (batched envs, actions already on the device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import humanoid_gym_amd as hg

for env_id, n_envs, kw in (("nao-v0", 4096, {}), ("pepper-v0", 4096, {}), ("dancer-v0", 4096, {}), ("dancer-v0", 4096, {"auto_reset": True})):
    env = hg.make(env_id, n_envs=n_envs, **kw)
    a = torch.tensor(np.tile(env.init_angles, (n_envs, 1)), dtype=torch.float32, device="cuda:0")
    for _ in range(20):
        env.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 300
    for _ in range(K):
        env.step(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    # host-only cost: time to ENQUEUE a step (no sync)
    t0 = time.perf_counter()
    for _ in range(K):
        env.step(a)
    host = (time.perf_counter() - t0) / K
    torch.cuda.synchronize()
    print(f"{env_id:10s} {kw} n_envs={n_envs}: {dt * 1e6:7.1f} us/step end to end ({n_envs / dt / 1e6:.1f} M env-steps/s), "
          f"{host * 1e6:6.1f} us/step to enqueue", flush=True)
    env.close()
