"""Episode statistics: the only quantity that crosses GPUs.

Envs never read each other's state (dancer_env.py:17-53 touches one body), so
ranks own disjoint env slices and exchange nothing on the data path.  What a
trainer wants globally is a handful of sums; each rank keeps an 8-float vector
on its device ([n_env_steps, sum reward, n_done, n_collision, 0, 0, 0, 0]).
Bound to a sim, the per-step accumulation is fused into the step kernel (running
per-env counters) and ``hg_stats_flush`` folds them into the vector when a
reduction is due; ``hg_stats_accumulate`` / explicit tensors remain for callers
that compute reward and done themselves.  Every ``reduce_every`` steps the
vector is all-reduced (RCCL when the backend is nccl) on a side stream so that
the 64-byte, latency-bound collective never sits between two physics launches
(``side_stream=False`` keeps it on the caller's stream: a launch that fills the GPU,
like 4096 NAO envs, is displaced a little by any concurrent kernel -- measured on one
GPU, 16-step cadence: 121.6 us per step with the side stream, 120.4 without,
119.3 with no reduction; tools/stats_cadence.sh).
"""
from __future__ import annotations

from typing import Optional, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous env slice [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class EpisodeStats:
    N = 8

    def __init__(self, sim=None, reduce_every: int = 16, device=None, group=None, fused: bool = True,
                 side_stream: bool = True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.sim = sim
        self.fused = bool(fused and sim is not None)    # per-step accumulation lives in the step kernel
        self.device = sim.device if sim is not None else torch.device(device or "cpu")
        self.local = torch.zeros(self.N, dtype=torch.float32, device=self.device)
        self.total = torch.zeros(self.N, dtype=torch.float32, device=self.device)   # global, as of last reduce
        self.reduce_every = int(reduce_every)
        self.group = group
        # a one-rank group still goes through the collective, so that the path can be rehearsed on one GPU
        self.distributed = dist.is_available() and dist.is_initialized()
        self.device_collective = self.distributed and (self.device.type != "cuda" or dist.get_backend(group) == "nccl")
        self.side = torch.cuda.Stream(self.device) if (self.device.type == "cuda" and side_stream) else None
        self._pending = None
        self._local_dirty = False       # something was added to `local` by accumulate() since the last reduction

    # -- accumulation --------------------------------------------------
    def accumulate(self, reward=None, done=None, collision=None):
        """Adds one step's per-env tensors.  Uses the HIP reduction when bound to a sim."""
        self._local_dirty = True
        if self.sim is not None and reward is None:
            self.sim.stats_accumulate(self.local)
            return
        t = self.torch
        n = (reward if reward is not None else done if done is not None else collision).numel()
        add = t.zeros(self.N, dtype=t.float32, device=self.device)
        add[0] = n
        if reward is not None:
            add[1] = reward.float().sum()
        if done is not None:
            add[2] = (done != 0).float().sum()
        if collision is not None:
            add[3] = (collision != 0).float().sum()
        self.local += add

    def update(self, step_index: int):
        if not self.fused:
            self.accumulate()
        if self.reduce_every > 0 and (step_index + 1) % self.reduce_every == 0:
            self.reduce_async()

    # -- reduction -----------------------------------------------------
    def reduce_async(self):
        """Snapshot the local sums and all-reduce the snapshot off the compute stream."""
        t = self.torch
        self._wait()
        if not self.distributed:
            # one rank: the counters go straight into the running total (one small launch instead of four)
            if self.fused:
                self.sim.stats_flush(self.total)
            if self._local_dirty:
                self.total += self.local
                self.local.zero_()
                self._local_dirty = False
            return
        if self.fused:
            self.sim.stats_flush(self.local)
        snap = self.local.clone()
        self.local.zero_()
        if not self.device_collective:
            # gloo with device tensors (CPU rehearsal of the multi-GPU path only): stage through the host
            host = snap.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            self.total += host.to(self.device)
            return
        if self.side is not None:
            self.side.wait_stream(t.cuda.current_stream(self.device))
            with t.cuda.stream(self.side):
                self.dist.all_reduce(snap, op=self.dist.ReduceOp.SUM, group=self.group)
            snap.record_stream(self.side)
            self._pending = snap
        else:
            self.dist.all_reduce(snap, op=self.dist.ReduceOp.SUM, group=self.group)
            self.total += snap

    def _wait(self):
        if self._pending is not None:
            self.torch.cuda.current_stream(self.device).wait_stream(self.side)
            self.total += self._pending
            self._pending = None

    def flush(self):
        self.reduce_async()
        self._wait()
        return self.total

    def as_dict(self):
        v = self.total.tolist()
        return {"env_steps": v[0], "reward_sum": v[1], "done": v[2], "collision": v[3]}
