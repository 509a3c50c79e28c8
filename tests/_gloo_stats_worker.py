"""Worker of test_stats_allreduce_world2_gloo (launched by torch.distributed.run, CPU, gloo)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from humanoid_gym_amd.stats import EpisodeStats, shard_range  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = shard_range(10, rank, world)          # rank 0: envs 0..4, rank 1: envs 5..9
st = EpisodeStats(None, reduce_every=2, device="cpu")
assert st.distributed
for t in range(4):
    env_ids = torch.arange(lo, hi)
    st.accumulate(reward=env_ids.float(), done=(env_ids % 5 == 0), collision=(env_ids % 2 == 0))
    if (t + 1) % 2 == 0:
        st.reduce_async()
tot = st.flush().tolist()
assert tot[0] == 40.0 and tot[1] == 4 * 45.0 and tot[2] == 4 * 2 and tot[3] == 4 * 5, tot
dist.barrier()
if rank == 0:
    print("GLOO_STATS_OK", tot[:4])
dist.destroy_process_group()
